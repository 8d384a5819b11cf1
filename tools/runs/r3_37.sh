#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for i in 1 2 3; do
for V in "" bnb4; do
if [ -z "$V" ]; then L=""; else L=$PWD/gpurun_exp_$V.so; fi
CRIMAC_FOLD_BNFIN=0 CRIMAC_LIB=$L timeout -k 10 200 python bench.py --precision h3p $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h3p nofold lib=$V', d['value'], d['ms_per_step'])"
done; done
