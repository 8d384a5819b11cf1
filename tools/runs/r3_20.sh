#!/bin/bash
set -e
mkdir -p gpurun_out/r3_20
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_h3p.py tests/test_gpu_unet.py tests/test_gpu_lowp_layerwise.py tests/test_lmi.py -q -x -m gpu > gpurun_out/r3_20/tests.log 2>&1 || { tail -60 gpurun_out/r3_20/tests.log; exit 1; }
tail -2 gpurun_out/r3_20/tests.log
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for P in bf16 h3p; do
for W in 1 0 1 0; do
CRIMAC_FOLD_BNFIN=$W timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P fold=$W', d['value'], d['ms_per_step'])"
done; done
