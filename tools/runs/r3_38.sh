#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for i in 1 2 3; do
for P in h3p bf16; do
for F in 1 0; do
CRIMAC_FOLD_BNFIN=$F timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P fold=$F', d['value'], d['ms_per_step'])"
done; done; done
