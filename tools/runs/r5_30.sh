#!/bin/bash
# grid-stride streaming kernels launched as whole resident rounds (shipped) vs the 2048-workgroup cap (CRIMAC_WHOLE_ROUNDS=0):
# kernel tests, timed step, serialized per-kernel totals (rocprofv3 kernel trace)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_30; mkdir -p $R
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -x -q > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log | cut -c1-250; exit 1; }
tail -2 $R/pytest.log
BARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop"
for S in 0 1 0 1; do
  CRIMAC_WHOLE_ROUNDS=$S timeout -k 10 200 python bench.py $BARGS > $R/bench_$S.json 2> $R/bench_$S.err || { tail $R/bench_$S.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$S.json')); print('whole_rounds=$S', round(d['ms_per_step'],3), 'ms', round(d['infer_patches_per_s']), 'infer')"
done
PARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide --no-train-loop --roofline-steps 3 --roofline-warmup 1"
for S in 0 1; do
  cd /tmp
  CRIMAC_WHOLE_ROUNDS=$S CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$S -- python3 $GRAFT_REPO_ROOT/bench.py $PARGS > $R/prof_$S.log 2>&1 || { echo prof failed; tail -20 $R/prof_$S.log; exit 1; }
  cd $GRAFT_REPO_ROOT
  t=$(find $R/prof_$S -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py $t > $R/step_breakdown_$S.txt
  rm -rf $R/prof_$S
  echo "== whole_rounds=$S"; head -24 $R/step_breakdown_$S.txt
done
echo r5_30 done
