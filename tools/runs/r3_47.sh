#!/bin/bash
set -e
mkdir -p gpurun_out/r3_47
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_h3p.py -q -x -m gpu -k "conv3x3 or network or first_layer or upconv" > gpurun_out/r3_47/t1.log 2>&1 || { tail -60 gpurun_out/r3_47/t1.log; exit 1; }
tail -1 gpurun_out/r3_47/t1.log
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide"
for i in 1 2 3; do
for P in bf16 h3p; do
for V in "" base; do
if [ -z "$V" ]; then L=""; else L=$PWD/gpurun_exp_$V.so; fi
CRIMAC_LIB=$L timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P lib=$V', d['value'], d['ms_per_step'], d['infer_patches_per_s'], d['roofline']['frac'])"
done; done; done
