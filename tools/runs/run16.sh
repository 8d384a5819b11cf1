#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "conv3x3 or fused_maxpool or cols" > gpurun_out/r16_tests.log 2>&1 || { tail -30 gpurun_out/r16_tests.log; exit 1; }
tail -2 gpurun_out/r16_tests.log
for b in 32 96; do
  echo "== new B=$b"; timeout -k 10 120 python tools/bench_p64.py $b
  echo "== old B=$b"; CRIMAC_LIB=$PWD/gpurun_exp_p64old.so timeout -k 10 120 python tools/bench_p64.py $b
done
CRIMAC_LIB=$PWD/gpurun_exp_diagph.so timeout -k 10 300 python tools/diag_p64_phases.py 32
