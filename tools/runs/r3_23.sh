#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r3_23
mkdir -p $R
export TMPDIR=/tmp
P=${1:-h3p}
BARGS="--precision $P --steps 4 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
cd /tmp
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/p -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/p.log 2>&1 || { echo prof failed; tail -20 $R/p.log; exit 1; }
t=$(find $R/p -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/step_breakdown.py $t 100 > $R/${P}_breakdown.txt
rm -rf $R/p
cat $R/${P}_breakdown.txt
