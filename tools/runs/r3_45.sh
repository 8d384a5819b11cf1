#!/bin/bash
set -e
for v in diagconv nostage; do
echo "== $v"
CRIMAC_LIB=$PWD/gpurun_exp_$v.so timeout -k 10 300 python tools/diag_wch_phases.py h3p 2>&1 | grep -v amdgpu | head -5
CRIMAC_LIB=$PWD/gpurun_exp_$v.so timeout -k 10 300 python tools/diag_wch_phases.py bf16 2>&1 | grep -v amdgpu | head -3
done
