#!/bin/bash
# A/B inside one call: bf16 (and fp16) step with the conv3x3 weight gradients grouped per gradient range
# (crimac_wgrad_group) against one launch per layer, and the plan's items per layer / layers per launch.
cd "$(dirname "$0")/../.." || exit 1
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --no-wide --no-train-loop --no-infer --steps 40 --warmup 10 --roofline-steps 12 --roofline-warmup 4"
run() {   # tag, env...
  tag=$1; shift
  env "$@" $B > gpurun_out/r4_02_$tag.json 2> gpurun_out/r4_02_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/r4_02_$tag.err; exit 1; }
}
for i in 1 2; do
  run off_$i CRIMAC_WGRAD_GROUP=0
  run g256_$i CRIMAC_WGRAD_GROUP=1
  run g128_$i CRIMAC_WGRAD_GROUP=1 CRIMAC_WGRAD_GROUP_ITEMS=128
  run g192_$i CRIMAC_WGRAD_GROUP=1 CRIMAC_WGRAD_GROUP_ITEMS=192
  run g384_$i CRIMAC_WGRAD_GROUP=1 CRIMAC_WGRAD_GROUP_ITEMS=384
  run l4_$i CRIMAC_WGRAD_GROUP=1 CRIMAC_WGRAD_GROUP_LAYERS=4
  run l2_$i CRIMAC_WGRAD_GROUP=1 CRIMAC_WGRAD_GROUP_LAYERS=2
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_02_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    w, c = d["roofline_wgrad"], d["roofline"]
    print(f"{f[21:-5]:10s} step {d['ms_per_step']:.3f} ms  wgrad: {w['launches_per_step']} launches, serial sum {w['median_launch_us'] * w['launches_per_step'] / 1e3:.3f} ms, frac {w['frac']:.3f}"
          f"  conv frac {c['frac']:.3f}  loss {d['final_loss']:.4f}")
PY
