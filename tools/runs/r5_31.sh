#!/bin/bash
# streaming BatchNorm passes walking their rows from the END of the tensor (what the producer wrote last): step A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_31; mkdir -p $R
BARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop --no-infer"
for V in base reverse base reverse; do
  if [ "$V" = base ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_$V.so; fi
  timeout -k 10 200 python bench.py $BARGS > $R/bench_$V.json 2> $R/bench_$V.err || { tail $R/bench_$V.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$V.json')); print('$V', round(d['ms_per_step'],3), 'ms')"
done
echo r5_31 done
