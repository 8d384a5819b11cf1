#!/bin/bash
# the weight stream of conv3x3_wch_kernel: lane-consecutive (coalesced) loads as an ablation, cache policies nt / sc1 / sc0
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_08; mkdir -p $R
for L in base coal nt sc1 sc0 base; do
  if [ "$L" = base ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_wch$L.so; fi
  timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$L.txt 2>&1 || { tail $R/launches_$L.txt; exit 1; }
  echo "variant $L: conv $(grep crimac_conv3x3 $R/launches_$L.txt | awk '{s+=$6} END {print s}') us; $(tail -1 $R/launches_$L.txt)"
done
paste <(grep crimac_conv3x3 $R/launches_base.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_coal.txt | awk '{print $6}') <(grep crimac_conv3x3 $R/launches_nt.txt | awk '{print $6}') <(grep crimac_conv3x3 $R/launches_sc1.txt | awk '{print $6}')
echo r5_08 done
