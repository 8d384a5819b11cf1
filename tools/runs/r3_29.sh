#!/bin/bash
set -e
mkdir -p gpurun_out/r3_29
timeout -k 10 1100 python -m pytest tests -q -x -m gpu > gpurun_out/r3_29/tests.log 2>&1 || { tail -60 gpurun_out/r3_29/tests.log; exit 1; }
tail -2 gpurun_out/r3_29/tests.log
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide"
for i in 1 2; do
for P in bf16 h3p; do
timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P', d['value'], d['ms_per_step'], d['infer_patches_per_s'], d['roofline']['frac'], d['roofline_wgrad']['frac'])"
done; done
