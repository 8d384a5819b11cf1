#!/bin/bash
# h3p: serialized kernel trace of a training step + inference
set -e
mkdir -p gpurun_out/r3_06
export TMPDIR=/tmp
CRIMAC_WGRAD_STREAM=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_06/prof -o h3p --output-format csv -- \
  python3 bench.py --precision h3p --no-parity-mode --no-tiled --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r3_06/bench.json 2> gpurun_out/r3_06/bench.err || { tail -20 gpurun_out/r3_06/bench.err; exit 1; }
tail -c 400 gpurun_out/r3_06/bench.json
f=$(find gpurun_out/r3_06/prof -name "*kernel_trace.csv" | head -1)
python3 tools/step_breakdown.py $f 100000 > gpurun_out/r3_06/breakdown.txt
tail -30 gpurun_out/r3_06/breakdown.txt
