#!/bin/bash
# items per layer of the grouped weight-gradient launch: 64 / 96 / 128 (default) on the bf16 and h3f steps, same box
cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out/r4_07; mkdir -p $R
A="--no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide --no-train-loop --roofline-steps 3 --roofline-warmup 1 --steps 30"
for i in 1 2; do
  for P in bf16 h3f; do
    for N in 128 64 96; do
      CRIMAC_WGRAD_GROUP_ITEMS=$N timeout -k 10 200 python bench.py --precision $P $A > $R/${P}_n${N}_$i.json 2> $R/${P}_n${N}_$i.err || exit 1
      python -c "import json,sys; d=json.load(open('$R/${P}_n${N}_$i.json')); print('$P items=$N run $i: %.3f ms/step, wgrad frac %.3f' % (d['ms_per_step'], d['roofline_wgrad']['frac']))"
    done
  done
done
