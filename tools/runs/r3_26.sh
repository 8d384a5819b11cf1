#!/bin/bash
set -e
mkdir -p gpurun_out/r3_26
timeout -k 10 600 python -m pytest tests/test_gpu_h3p.py -q -x -m gpu > gpurun_out/r3_26/t1.log 2>&1 || { tail -60 gpurun_out/r3_26/t1.log; exit 1; }
tail -1 gpurun_out/r3_26/t1.log
CRIMAC_CONV_TALL16=2 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "conv3x3" > gpurun_out/r3_26/t2.log 2>&1 || { tail -60 gpurun_out/r3_26/t2.log; exit 1; }
tail -1 gpurun_out/r3_26/t2.log
for v in 2 1 2 1; do
echo "== h3p CRIMAC_CONV_S22=$v"
CRIMAC_CONV_S22=$v timeout -k 10 200 python tools/bench_conv.py conv --prec h3p --iters 20 2>&1 | grep -E "e0c2|d3c1|total"
done
for v in 0 1 2 0 2; do
echo "== bf16 CRIMAC_CONV_TALL16=$v"
CRIMAC_CONV_TALL16=$v timeout -k 10 200 python tools/bench_conv.py conv --prec bf16 --iters 20 2>&1 | grep -E "e0c2|d3c1|total"
done
