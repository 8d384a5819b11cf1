#!/bin/bash
# p64 with the operand swap (shipped) vs before (gpurun_exp_p64old.so): per-launch tables, three alternations
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_40; mkdir -p $R
for V in old new old new old new; do
  if [ "$V" = new ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_p64old.so; fi
  timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$V.txt 2>&1 || { tail $R/launches_$V.txt; exit 1; }
  echo "$V $(tail -1 $R/launches_$V.txt) conv: $(grep crimac_conv3x3 $R/launches_$V.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_old.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_new.txt | awk '{print $6}') | awk '$1==0 || $1==1 || $1==21 || $1==22 || $1==23 || $1==24 || $1==50'
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "c16 or first or cin or stat or cols" 2>&1 | tail -2
echo r5_40 done
