#!/bin/bash
# one rank: the encoder's (and decoder's) weight-gradient groups in fewer persistent launches (CRIMAC_WGRAD_MERGE = digits
# of the backward groups whose hand-over is deferred to the next group's) -- step time A/B and the whole-net gradient tests
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_21; mkdir -p $R
BARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop --no-infer"
for V in off 1 12 012 2 off 1 12 012 2; do
  if [ "$V" = off ]; then unset CRIMAC_WGRAD_MERGE; else export CRIMAC_WGRAD_MERGE=$V; fi
  timeout -k 10 200 python bench.py $BARGS > $R/bench_$V.json 2> $R/bench_$V.err || { tail $R/bench_$V.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$V.json')); print('merge=$V', round(d['ms_per_step'],3), 'ms', 'wgrad frac', d.get('roofline_wgrad',{}).get('frac'))"
done
export CRIMAC_WGRAD_MERGE=12
timeout -k 10 400 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "golden or full_step or train_step" > $R/tests.log 2>&1; tail -3 $R/tests.log
echo r5_21 done
