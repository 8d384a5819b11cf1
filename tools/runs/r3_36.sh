#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for i in 1 2 3; do
for V in "" bnb2; do
if [ -z "$V" ]; then L=""; else L=$PWD/gpurun_exp_$V.so; fi
for P in h3p bf16; do
CRIMAC_LIB=$L timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P lib=$V', d['value'], d['ms_per_step'])"
done; done; done
for V in "" bnb2; do
if [ -z "$V" ]; then L=""; else L=$PWD/gpurun_exp_$V.so; fi
CRIMAC_WGRAD_STREAM=0 CRIMAC_LIB=$L timeout -k 10 200 python bench.py --precision h3p $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h3p serialized lib=$V', d['value'], d['ms_per_step'])"
done
