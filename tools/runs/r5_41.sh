#!/bin/bash
# inference: the batch as two halves on two streams (shipped) vs one stream (CRIMAC_EVAL_STREAMS=0), bf16 and h3p
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_41; mkdir -p $R
for P in bf16 h3p; do
  for S in 1 0 1 0; do
    CRIMAC_EVAL_STREAMS=$S timeout -k 10 200 python bench.py --precision $P --steps 6 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop --roofline-steps 2 --roofline-warmup 1 > $R/bench_${P}_$S.json 2> $R/bench_${P}_$S.err || { tail $R/bench_${P}_$S.err; exit 1; }
    python -c "
import json; d=json.load(open('$R/bench_${P}_$S.json')); print('$P streams=$S', round(d['infer_patches_per_s']), 'infer patches/s')"
  done
done
echo r5_41 done
