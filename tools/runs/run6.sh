#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2f
timeout -k 10 600 python -m pytest tests/test_lmi.py -m gpu -q -x -s > gpurun_out/r2f/tests.log 2>&1
rc=$?
tail -25 gpurun_out/r2f/tests.log
exit $rc
