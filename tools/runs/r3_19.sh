#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for P in bf16 h3p; do
for W in 512 256 384 512 256; do
CRIMAC_WGRAD_BLOCKS=$W timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P blocks=$W', d['value'], d['ms_per_step'], d['roofline_wgrad']['frac'])"
done; done
