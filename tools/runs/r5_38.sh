#!/bin/bash
# single weight-gradient launches (first layer + four transposed convolutions): workgroups per launch (CRIMAC_WGRAD_BLOCKS)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_38; mkdir -p $R
for S in 512 256 1024 768 512 256; do
  CRIMAC_WGRAD_BLOCKS=$S timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$S.txt 2>&1 || { tail $R/launches_$S.txt; exit 1; }
  echo "blocks=$S $(tail -1 $R/launches_$S.txt) wgrad singles: $(grep 'crimac_wgrad ' $R/launches_$S.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep "crimac_wgrad " $R/launches_512.txt | awk '{print $1, $3, $6}') <(grep "crimac_wgrad " $R/launches_256.txt | awk '{print $6}') <(grep "crimac_wgrad " $R/launches_768.txt | awk '{print $6}') <(grep "crimac_wgrad " $R/launches_1024.txt | awk '{print $6}')
echo r5_38 done
