#!/bin/bash
# end of round 5: the whole GPU suite, then the default bench twice (profiles/r05_bench_{a,b}.json)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_15; mkdir -p $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1 || { tail -40 $R/pytest_gpu.log; exit 1; }
tail -3 $R/pytest_gpu.log
timeout -k 10 600 python bench.py > $R/bench_b.json 2> $R/bench_b.err || { tail -30 $R/bench_b.err; exit 1; }
grep "timed region\|train_loop\|tiled\|wide" $R/bench_b.err
echo r5_15 done
