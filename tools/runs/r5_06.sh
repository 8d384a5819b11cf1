#!/bin/bash
# where the <= 2-chunk launches of conv3x3_wch_kernel lose their time: phase stamps (bf16) and the no-weight-stream ablation
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_06; mkdir -p $R
CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_wchphases.so timeout -k 10 300 python tools/diag_wch_phases.py bf16 > $R/phases_bf16.txt 2>&1 || { tail $R/phases_bf16.txt; exit 1; }
cat $R/phases_bf16.txt
for L in "" now; do
  if [ -n "$L" ]; then export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_wch$L.so; fi
  timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$L.txt 2>&1 || { tail $R/launches_$L.txt; exit 1; }
  echo "variant '$L': conv $(grep crimac_conv3x3 $R/launches_$L.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_now.txt | awk '{print $6}')
echo r5_06 done
