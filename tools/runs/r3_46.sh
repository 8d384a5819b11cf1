#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide"
for i in 1 2 3; do
for P in bf16 h3p; do
for V in "" nobiaspre; do
if [ -z "$V" ]; then L=""; else L=$PWD/gpurun_exp_$V.so; fi
CRIMAC_LIB=$L timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P lib=$V', d['value'], d['ms_per_step'], d['infer_patches_per_s'], d['roofline']['frac'])"
done; done; done
