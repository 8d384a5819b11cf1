#!/bin/bash
# weight stream of the channel-split convolution: every wave streams the SAME channel block (L1 hits) against the real stream
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_17; mkdir -p $R
for L in base samew halfw base samew; do
  if [ "$L" = base ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_wch$L.so; fi
  timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$L.txt 2>&1 || { tail $R/launches_$L.txt; exit 1; }
  echo "variant $L: conv $(grep crimac_conv3x3 $R/launches_$L.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_base.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_samew.txt | awk '{print $6}') <(grep crimac_conv3x3 $R/launches_halfw.txt | awk '{print $6}')
echo r5_17 done
