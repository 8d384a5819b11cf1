#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for r in 1 2; do
for P in bf16 h3p; do
for W in 0 1; do
CRIMAC_WGRAD_PARTIALS=$W timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P partials=$W', d['value'], d['ms_per_step'], d['roofline_wgrad']['frac'])"
done; done; done
