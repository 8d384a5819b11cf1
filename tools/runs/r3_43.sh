#!/bin/bash
set -e
mkdir -p gpurun_out/r3_43
timeout -k 10 900 python -m pytest tests/test_gpu_h3p.py tests/test_gpu_kernels.py -q -x -m gpu -k "upconv or network or igemm or transposed" > gpurun_out/r3_43/t1.log 2>&1 || { tail -60 gpurun_out/r3_43/t1.log; exit 1; }
tail -1 gpurun_out/r3_43/t1.log
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide"
for i in 1 2 3; do
for P in h3p bf16; do
for V in 1 0; do
CRIMAC_UPCONV_W8=$V timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P w8=$V', d['value'], d['ms_per_step'], d['infer_patches_per_s'])"
done; done; done
