#!/bin/bash
set -e
mkdir -p gpurun_out/r3_03
timeout -k 10 600 python tools/check_h3p.py golden timing > gpurun_out/r3_03/check.log 2>&1 || { tail -40 gpurun_out/r3_03/check.log; exit 1; }
cat gpurun_out/r3_03/check.log | grep -v "^$" | tail -60
