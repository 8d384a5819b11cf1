#!/bin/bash
# channel-split convolution, what the LDS fragment reads cost: ablation builds with 3 of 8 read groups (the read volume of a
# form that uses each halo-row fragment for the three taps of its column), without the weight stream, and both
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_22; mkdir -p $R
for V in base fewa now fewa_now base fewa; do
  if [ "$V" = base ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_$V.so; fi
  timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$V.txt 2>&1 || { tail $R/launches_$V.txt; exit 1; }
  echo "$V $(tail -1 $R/launches_$V.txt) conv: $(grep crimac_conv3x3 $R/launches_$V.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_base.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_fewa.txt | awk '{print $6}') <(grep crimac_conv3x3 $R/launches_now.txt | awk '{print $6}') <(grep crimac_conv3x3 $R/launches_fewa_now.txt | awk '{print $6}')
echo r5_22 done
