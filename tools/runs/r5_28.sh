#!/bin/bash
# whole GPU suite + smoke + the default bench on the tree with the rows form in the library (profiles/r05_bench_c.json)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_28; mkdir -p $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1 || { tail -40 $R/pytest_gpu.log | cut -c1-300; exit 1; }
tail -3 $R/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1 || { tail -20 $R/smoke.log; exit 1; }
tail -2 $R/smoke.log
timeout -k 10 600 python bench.py > $R/bench_c.json 2> $R/bench_c.err || { tail -30 $R/bench_c.err; exit 1; }
grep "timed region\|train_loop\|tiled\|wide" $R/bench_c.err
echo r5_28 done
