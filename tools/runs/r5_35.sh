#!/bin/bash
# weighted cross-entropy forward with 16-byte loads and at most 512 workgroups: loss tests, serialized per-kernel totals
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_35; mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -x -q -k "wce or loss or golden or ignored or pool" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log | cut -c1-250; exit 1; }
tail -2 $R/pytest.log
PARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide --no-train-loop --roofline-steps 3 --roofline-warmup 1"
cd /tmp
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -- python3 $GRAFT_REPO_ROOT/bench.py $PARGS > $R/prof.log 2>&1 || { echo prof failed; tail -20 $R/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
t=$(find $R/prof -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py $t > $R/step_breakdown.txt
rm -rf $R/prof
head -26 $R/step_breakdown.txt
echo r5_35 done
