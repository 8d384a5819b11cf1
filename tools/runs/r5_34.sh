#!/bin/bash
# transposed convolutions: 256 columns per workgroup (CRIMAC_UPCONV_W8=1, one 8-wave workgroup per CU) vs 128, per launch
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_34; mkdir -p $R
for S in 0 1 0 1; do
  CRIMAC_UPCONV_W8=$S timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$S.txt 2>&1 || { tail $R/launches_$S.txt; exit 1; }
  echo "w8=$S $(tail -1 $R/launches_$S.txt)"
done
paste <(grep "igemm\|upconv" $R/launches_0.txt | awk '{print $1, $2, $6}') <(grep "igemm\|upconv" $R/launches_1.txt | awk '{print $6}')
echo r5_34 done
