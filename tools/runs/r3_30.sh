#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for i in 1 2 3; do
for V in 2 1; do
CRIMAC_CONV_S22=$V timeout -k 10 200 python bench.py --precision h3p $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h3p S22=$V', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
for V in 1 0; do
CRIMAC_CONV_TALL16=$V timeout -k 10 200 python bench.py --precision bf16 $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 TALL16=$V', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
done
