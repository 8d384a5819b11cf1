#!/bin/bash
# GPU tests of the files the -x run of r5_02 did not reach + the new robustness tests (with their printed tables)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_04; mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_labels.py tests/test_lmi.py tests/test_tiling.py tests/test_worker_chain.py tests/test_gpu_parity_robustness.py -m gpu -q -s > $R/pytest_gpu.log 2>&1; rc=$?
grep -v "^\[build\]" $R/pytest_gpu.log | tail -60
exit $rc
