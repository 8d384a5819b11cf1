#!/bin/bash
# A/B of CRIMAC_FUSE_UNPOOL_APPLY (unpool + BatchNorm-backward apply in one pass) on the bf16 and h3f steps, same box
cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out/r4_06; mkdir -p $R
A="--no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide --no-train-loop --roofline-steps 3 --roofline-warmup 1 --steps 30"
for i in 1 2; do
  for P in bf16 h3f; do
    for F in 0 1; do
      CRIMAC_FUSE_UNPOOL_APPLY=$F timeout -k 10 200 python bench.py --precision $P $A > $R/${P}_f${F}_$i.json 2> $R/${P}_f${F}_$i.err || exit 1
      python -c "import json,sys; d=json.load(open('$R/${P}_f${F}_$i.json')); print('$P fuse=$F run $i: %.3f ms/step' % d['ms_per_step'])"
    done
  done
done
