#!/bin/bash
set -e
mkdir -p gpurun_out/r3_31
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_h3p.py tests/test_gpu_lowp_layerwise.py -q -x -m gpu > gpurun_out/r3_31/t1.log 2>&1 || { tail -60 gpurun_out/r3_31/t1.log; exit 1; }
tail -1 gpurun_out/r3_31/t1.log
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for i in 1 2 3; do
for V in "" bnbinline; do
if [ -z "$V" ]; then L=""; else L=$PWD/gpurun_exp_$V.so; fi
for P in h3p bf16; do
CRIMAC_LIB=$L timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P lib=$V', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done; done
