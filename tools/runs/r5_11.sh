#!/bin/bash
# rocprofv3 kernel stats of the default bench command without the train_loop legs (DataLoader workers are not forked under the profiler), for profiles/
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_11
mkdir -p $R
export TMPDIR=/tmp
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/p -- python3 $GRAFT_REPO_ROOT/bench.py --no-train-loop > $R/bench.json 2> $R/bench.err || { echo failed; tail -20 $R/bench.err; exit 1; }
f=$(find $R/p -name "*kernel_stats.csv" | head -1); cp $f $R/default_bench_kernel_stats.csv
rm -rf $R/p
head -12 $R/default_bench_kernel_stats.csv | cut -c1-160
tail -1 $R/bench.json | cut -c1-300
