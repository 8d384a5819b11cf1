#!/bin/bash
# persistent 64 -> 64 kernel: what the staging arithmetic of its epilogue is worth (ablation build: one of four staging
# iterations per accumulator tile, results garbage) -- per-launch table of the bf16 step
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_37; mkdir -p $R
for V in base p64few base p64few; do
  if [ "$V" = base ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_$V.so; fi
  timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$V.txt 2>&1 || { tail $R/launches_$V.txt; exit 1; }
  echo "$V $(tail -1 $R/launches_$V.txt)"
done
paste <(grep crimac_conv3x3 $R/launches_base.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_p64few.txt | awk '{print $6}') | awk '$3 > 0 && ($4 < 0.97 * $3 || $4 > 1.03 * $3)'
echo r5_37 done
