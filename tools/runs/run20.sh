#!/bin/bash
set -e
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu > gpurun_out/r20_tests.log 2>&1 || { tail -30 gpurun_out/r20_tests.log; exit 1; }
tail -2 gpurun_out/r20_tests.log
CRIMAC_LIB=$PWD/gpurun_exp_diagcph.so timeout -k 10 300 python tools/diag_wch_phases.py 2>&1 | grep -v amdgpu
echo == new; timeout -k 10 200 python tools/bench_conv.py conv --iters 20 2>&1 | grep -v amdgpu
echo == old; CRIMAC_LIB=$PWD/gpurun_exp_p64old.so timeout -k 10 200 python tools/bench_conv.py conv --iters 20 2>&1 | grep -v amdgpu
