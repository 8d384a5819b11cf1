#!/bin/bash
# round 5, first call: train_loop_raw legs (bench.py) + per-launch table of the serialized bf16 / h3f step
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_01; mkdir -p $R
timeout -k 10 500 python bench.py --no-tiled --no-wide --no-cpu-baseline --no-infer > $R/bench_loops.json 2> $R/bench_loops.err || { echo bench failed; tail -30 $R/bench_loops.err; exit 1; }
grep "train_loop\|timed region" $R/bench_loops.err
timeout -k 10 200 python tools/step_launches.py bf16 30 > $R/bf16_serial_per_launch.txt 2>&1 || { echo launches failed; tail $R/bf16_serial_per_launch.txt; exit 1; }
timeout -k 10 200 python tools/step_launches.py h3f 20 > $R/h3f_serial_per_launch.txt 2>&1 || { echo launches failed; tail $R/h3f_serial_per_launch.txt; exit 1; }
tail -3 $R/bf16_serial_per_launch.txt
echo r5_01 done
