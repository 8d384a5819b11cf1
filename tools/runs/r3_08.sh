#!/bin/bash
set -e
mkdir -p gpurun_out/r3_08
timeout -k 10 600 python -m pytest tests/test_gpu_h3p.py -q -x -m gpu > gpurun_out/r3_08/tests.log 2>&1 || { tail -60 gpurun_out/r3_08/tests.log; exit 1; }
tail -2 gpurun_out/r3_08/tests.log
echo "== S22 on"; timeout -k 10 300 python tools/check_h3p.py timing 2>&1 | grep "h3p\]"
echo "== S22 off"; CRIMAC_CONV_S22=0 timeout -k 10 300 python tools/check_h3p.py timing 2>&1 | grep "h3p\]"
