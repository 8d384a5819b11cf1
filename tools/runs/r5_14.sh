#!/bin/bash
# what the weight gradients cost the STEP (not the side stream): timed bf16 step with the single (non-grouped) launches
# skipped, with all weight gradients skipped, and as shipped
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_14; mkdir -p $R
BARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop --no-infer"
for V in none single all none single all; do
  if [ "$V" = none ]; then unset CRIMAC_EXP_SKIP_WGRAD; else export CRIMAC_EXP_SKIP_WGRAD=$V; fi
  timeout -k 10 200 python bench.py $BARGS > $R/bench_$V.json 2> $R/bench_$V.err || { tail $R/bench_$V.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$V.json')); print('skip=$V', round(d['ms_per_step'],3), 'ms')"
done
echo r5_14 done
