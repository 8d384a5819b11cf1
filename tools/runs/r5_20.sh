#!/bin/bash
# the tail of the GPU suite behind the last failure + the default bench (profiles/r05_bench_b.json)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_20; mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_gpu_unet.py tests/test_labels.py tests/test_lmi.py tests/test_tiling.py tests/test_worker_chain.py tests/test_gpu_parity_robustness.py -m gpu -x -q -k "row_major or labels or lmi or tiling or worker or parity or robust or structured or trained or golden or h3f" > $R/pytest_gpu.log 2>&1 || { tail -40 $R/pytest_gpu.log; exit 1; }
tail -3 $R/pytest_gpu.log
timeout -k 10 600 python bench.py > $R/bench_b.json 2> $R/bench_b.err || { tail -30 $R/bench_b.err; exit 1; }
grep "timed region\|train_loop\|tiled\|wide" $R/bench_b.err
echo r5_20 done
