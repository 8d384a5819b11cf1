#!/bin/bash
# clock the chip holds inside the convolution / weight-gradient kernels and the share of MFMA issue slots filled at that clock
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_25; mkdir -p $R
CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_diagconv.so timeout -k 10 300 python tools/diag_clock.py conv $R/clock_conv.json > $R/clock_conv.txt 2>&1 || { tail $R/clock_conv.txt; exit 1; }
cat $R/clock_conv.txt
CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_diagwgrad.so timeout -k 10 300 python tools/diag_clock.py wgrad $R/clock_wgrad.json > $R/clock_wgrad.txt 2>&1 || { tail $R/clock_wgrad.txt; exit 1; }
cat $R/clock_wgrad.txt
echo r5_25 done
