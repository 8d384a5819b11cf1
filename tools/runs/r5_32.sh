#!/bin/bash
# reverse traversal of the streaming passes: serialized per-kernel totals (rocprofv3 kernel trace) + alternating timed steps
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_32; mkdir -p $R
export TMPDIR=/tmp
PARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide --no-train-loop --roofline-steps 3 --roofline-warmup 1"
for V in base reverse; do
  if [ "$V" = base ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_$V.so; fi
  cd /tmp
  CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$V -- python3 $GRAFT_REPO_ROOT/bench.py $PARGS > $R/prof_$V.log 2>&1 || { echo prof failed; tail -20 $R/prof_$V.log; exit 1; }
  cd $GRAFT_REPO_ROOT
  t=$(find $R/prof_$V -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py $t > $R/step_breakdown_$V.txt
  rm -rf $R/prof_$V
  echo "== $V"; head -16 $R/step_breakdown_$V.txt
done
BARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop --no-infer"
for V in base reverse base reverse base reverse; do
  if [ "$V" = base ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_$V.so; fi
  timeout -k 10 200 python bench.py $BARGS > $R/bench_$V.json 2> $R/bench_$V.err || { tail $R/bench_$V.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$V.json')); print('$V', round(d['ms_per_step'],3), 'ms')"
done
echo r5_32 done
