#!/bin/bash
set -e
mkdir -p gpurun_out/r3_16
timeout -k 10 600 python -m pytest tests/test_gpu_h3p.py -q -x -m gpu -k "wgrad" > gpurun_out/r3_16/tests.log 2>&1 || { tail -40 gpurun_out/r3_16/tests.log; exit 1; }
tail -1 gpurun_out/r3_16/tests.log
for i in 1 2; do
echo "== DMA by waves 4-7"; timeout -k 10 200 python tools/bench_conv.py wgrad --prec h3p --iters 20 2>&1 | grep -v amdgpu | tail -14
echo "== DMA by all 8 waves"; CRIMAC_LIB=$PWD/gpurun_exp_ppdma8.so timeout -k 10 200 python tools/bench_conv.py wgrad --prec h3p --iters 20 2>&1 | grep -v amdgpu | tail -14
done
