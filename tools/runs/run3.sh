#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_gpu_lowp_layerwise.py tests/test_gpu_unet.py -m gpu -q -s -k "lowp or fp16 or layer or storage" > gpurun_out/r2c/tests.log 2>&1
rc=$?
grep -n "tensors checked\|storage-rounding\|passed\|failed\|Error\|^E " gpurun_out/r2c/tests.log | head -40
exit $rc
