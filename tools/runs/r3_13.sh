#!/bin/bash
set -e
for lib in "" "$PWD/gpurun_exp_nslot4.so"; do
  echo "=== lib: ${lib:-default}"
  for prec in bf16 h3p; do
    echo "--- $prec w4 (p64 / s22 off)"
    CRIMAC_LIB=$lib CRIMAC_CONV_P64=0 CRIMAC_CONV_S22=0 timeout -k 10 200 python tools/bench_conv.py conv --prec $prec --iters 20 --layers 0,12 2>&1 | grep -v amdgpu
  done
done
echo "--- h3p default dispatch (s22)"; timeout -k 10 200 python tools/bench_conv.py conv --prec h3p --iters 20 --layers 0,12 2>&1 | grep -v amdgpu
