#!/bin/bash
set -e
mkdir -p gpurun_out/r3_40
timeout -k 10 600 python -m pytest tests/test_gpu_h3p.py -q -x -m gpu -k "upconv" > gpurun_out/r3_40/t1.log 2>&1 || { tail -60 gpurun_out/r3_40/t1.log; exit 1; }
tail -1 gpurun_out/r3_40/t1.log
