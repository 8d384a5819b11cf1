#!/bin/bash
set -e
mkdir -p gpurun_out/r3_28
timeout -k 10 600 python -m pytest tests/test_gpu_h3p.py -q -x -m gpu > gpurun_out/r3_28/t1.log 2>&1 || { tail -60 gpurun_out/r3_28/t1.log; exit 1; }
tail -1 gpurun_out/r3_28/t1.log
CRIMAC_CONV_TALL16=2 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "conv3x3" > gpurun_out/r3_28/t2.log 2>&1 || { tail -60 gpurun_out/r3_28/t2.log; exit 1; }
tail -1 gpurun_out/r3_28/t2.log
CRIMAC_LIB=$PWD/gpurun_exp_diagconv.so timeout -k 10 300 python tools/diag_wch_phases.py h3p 2>&1 | grep -v amdgpu | head -3
for i in 1 2; do
timeout -k 10 200 python tools/bench_conv.py conv --prec h3p --iters 20 2>&1 | grep -E "e0c2|d3c1|total"
CRIMAC_CONV_TALL16=1 timeout -k 10 200 python tools/bench_conv.py conv --prec bf16 --iters 20 2>&1 | grep -E "e0c2|d3c1|total"
done
