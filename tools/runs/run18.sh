#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wgrad or partial" > gpurun_out/r18_tests.log 2>&1 || { tail -30 gpurun_out/r18_tests.log; exit 1; }
tail -2 gpurun_out/r18_tests.log
echo == two teams; timeout -k 10 200 python tools/bench_conv.py wgrad --iters 20 2>&1 | grep -v amdgpu
echo == one team; CRIMAC_WGRAD_TEAMS=1 timeout -k 10 200 python tools/bench_conv.py wgrad --iters 20 2>&1 | grep -v amdgpu
