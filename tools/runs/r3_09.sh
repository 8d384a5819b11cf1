#!/bin/bash
set -e
mkdir -p gpurun_out/r3_09
timeout -k 10 1100 python -m pytest tests -q -x -m gpu > gpurun_out/r3_09/tests.log 2>&1 || { tail -60 gpurun_out/r3_09/tests.log; exit 1; }
tail -3 gpurun_out/r3_09/tests.log
