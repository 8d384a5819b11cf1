#!/bin/bash
set -e
mkdir -p gpurun_out/r3_11
timeout -k 10 900 python -m pytest tests/test_gpu_h3p.py tests/test_tiling.py -q -x -m gpu > gpurun_out/r3_11/tests.log 2>&1 || { tail -60 gpurun_out/r3_11/tests.log; exit 1; }
tail -2 gpurun_out/r3_11/tests.log
timeout -k 10 300 python tools/check_h3p.py timing 2>&1 | grep "\]"
