#!/bin/bash
set -e
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
run() { # label, env...
  local label=$1; shift
  for P in bf16 h3p; do
  env "$@" timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P $label', round(d['value'],1), round(d['ms_per_step'],3))"
  done
}
run default X=1
run wgrad_stream=2 CRIMAC_WGRAD_STREAM=2
run split_skip=0 CRIMAC_SPLIT_SKIP=0
run unpack_side=0 CRIMAC_UNPACK_SIDE=0
run default X=1
run eval_streams=0 CRIMAC_EVAL_STREAMS=0
