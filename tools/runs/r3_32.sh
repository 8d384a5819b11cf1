#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r3_32
mkdir -p $R
export TMPDIR=/tmp
BARGS="--precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
cd /tmp
for W in 0 1; do
export CRIMAC_WGRAD_PARTIALS=$W
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/f$W -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/f$W.log 2>&1 || { echo prof failed; tail -20 $R/f$W.log; exit 1; }
t=$(find $R/f$W -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/step_breakdown.py $t > $R/f${W}_breakdown.txt
rm -rf $R/f$W
head -28 $R/f${W}_breakdown.txt
done
