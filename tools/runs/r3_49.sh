#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for i in 1 2 3 4; do
for V in 1 0; do
CRIMAC_SPLIT_SKIP=$V timeout -k 10 200 python bench.py --precision bf16 $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 split_skip=$V', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
