#!/bin/bash
# bn_bwd_apply (streaming form) at the network's shapes: rows in flight, waves per SIMD, grid size
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_29; mkdir -p $R
for V in base w2u8 w4u2 w4u4 w3u4; do
  if [ "$V" = base ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_$V.so; fi
  for G in 0 768 1024 1536; do
    echo "== $V grid cap $G"
    CRIMAC_BNB_GRID=$G timeout -k 10 100 python tools/bench_bnb.py 2>&1 | grep -v amdgpu || exit 1
  done
done > $R/bnb.txt 2>&1
grep "==\|per step\|C=   64" $R/bnb.txt
echo r5_29 done
