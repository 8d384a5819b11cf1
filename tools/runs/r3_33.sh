#!/bin/bash
set -e
for v in "" noatomic ""; do
echo "== variant '$v'"
if [ -z "$v" ]; then L=""; else L=$PWD/gpurun_exp_$v.so; fi
CRIMAC_LIB=$L timeout -k 10 200 python tools/bench_conv.py wgrad --prec bf16 --iters 20 2>&1 | grep -v amdgpu | tail -14
done
