#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r22
export TMPDIR=/tmp
cd /tmp
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r22/prof_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $GRAFT_REPO_ROOT/gpurun_out/r22/prof_serial.log 2>&1 || { echo prof serial failed; exit 1; }
cd $GRAFT_REPO_ROOT
t=$(find gpurun_out/r22/prof_serial -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py $t 60 > gpurun_out/r22/serial_step_breakdown.txt
rm -f $t
cat gpurun_out/r22/serial_step_breakdown.txt
