#!/bin/bash
# rows form vs 128-channel form of the channel-split kernel: phase stamps of a workgroup (wave 0), same shapes
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_24; mkdir -p $R
export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_wchphases.so
for F in 16 48; do
  timeout -k 10 300 python tools/diag_wch_phases.py bf16 $F > $R/phases_$F.txt 2>&1 || { tail $R/phases_$F.txt; exit 1; }
  echo "flags=$F"; cat $R/phases_$F.txt
done
echo r5_24 done
