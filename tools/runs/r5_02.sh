#!/bin/bash
# full GPU test suite (round 5: ABI v6)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_02; mkdir -p $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; rc=$?
tail -15 $R/pytest_gpu.log
exit $rc
