#!/bin/bash
# round 3, run 1: where does the parity-precision (f32h3) step spend its time?  serialized kernel trace.
set -e
mkdir -p gpurun_out/r3_01
export TMPDIR=/tmp
CRIMAC_WGRAD_STREAM=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_01/prof -o f32h3 --output-format csv -- \
  python3 bench.py --precision f32h3 --no-parity-mode --no-tiled --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r3_01/bench.json 2> gpurun_out/r3_01/bench.err
tail -c 600 gpurun_out/r3_01/bench.json
f=$(find gpurun_out/r3_01/prof -name "*kernel_trace.csv" | head -1)
python3 tools/step_breakdown.py $f 150 > gpurun_out/r3_01/breakdown.txt
cat gpurun_out/r3_01/breakdown.txt | tail -40
