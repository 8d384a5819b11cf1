#!/bin/bash
# bf16: the fused unpool + BatchNorm-backward apply path (CRIMAC_FUSE_UNPOOL_APPLY=1) re-measured with the faster sums-only pass
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_13; mkdir -p $R
BARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop --no-infer"
for F in 0 1 0 1 0 1; do
  CRIMAC_FUSE_UNPOOL_APPLY=$F timeout -k 10 200 python bench.py $BARGS > $R/bench_$F.json 2> $R/bench_$F.err || { tail $R/bench_$F.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$F.json')); print('fuse_unpool_apply=$F', round(d['ms_per_step'],3), 'ms')"
done
echo r5_13 done
