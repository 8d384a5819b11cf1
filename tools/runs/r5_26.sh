#!/bin/bash
# (a) statistics fold of the convolution epilogue behind the stores (shipped) vs in front of them (gpurun_exp_foldfirst.so);
# (b) rows form for the 64-output-channel launches only (CRIMAC_CONV_ROWS=64), bf16 and h3f: per-launch tables and step
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_26; mkdir -p $R
run() {   # tag, precision
  timeout -k 10 200 python tools/step_launches.py $2 20 > $R/launches_$1_$2.txt 2>&1 || { tail $R/launches_$1_$2.txt; exit 1; }
  echo "$1 $2 $(tail -1 $R/launches_$1_$2.txt) conv: $(grep crimac_conv3x3 $R/launches_$1_$2.txt | awk '{s+=$6} END {print s}') us"
}
for rep in 1 2; do
  unset CRIMAC_LIB CRIMAC_CONV_ROWS; run new bf16
  export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_foldfirst.so; run old bf16; unset CRIMAC_LIB
  export CRIMAC_CONV_ROWS=64; run rows64 bf16; unset CRIMAC_CONV_ROWS
done
paste <(grep crimac_conv3x3 $R/launches_new_bf16.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_old_bf16.txt | awk '{print $6}') <(grep crimac_conv3x3 $R/launches_rows64_bf16.txt | awk '{print $6}')
for rep in 1 2; do
  unset CRIMAC_CONV_ROWS; run new h3f
  export CRIMAC_CONV_ROWS=64; run rows64 h3f; unset CRIMAC_CONV_ROWS
done
paste <(grep crimac_conv3x3 $R/launches_new_h3f.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_rows64_h3f.txt | awk '{print $6}')
echo r5_26 done
