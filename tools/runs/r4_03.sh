#!/bin/bash
# A/B inside one call, plane-pair precision (h3p): conv3x3 weight gradients grouped per gradient range vs one launch per layer.
cd "$(dirname "$0")/../.." || exit 1
B="python bench.py --precision h3p --no-cpu-baseline --no-parity-mode --no-tiled --no-wide --no-train-loop --no-infer --steps 20 --warmup 5 --roofline-steps 8 --roofline-warmup 3"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r4_03_$tag.json 2> gpurun_out/r4_03_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/r4_03_$tag.err; exit 1; }; }
for i in 1 2; do
  run off_$i CRIMAC_WGRAD_GROUP=0
  run g128_$i CRIMAC_WGRAD_GROUP=1
  run g256_$i CRIMAC_WGRAD_GROUP=1 CRIMAC_WGRAD_GROUP_ITEMS=256
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r4_03_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    w, c = d["roofline_wgrad"], d["roofline"]
    print(f"{os.path.basename(f)[6:-5]:8s} step {d['ms_per_step']:.3f} ms  wgrad: {w['launches_per_step']} launches, serial sum {w['median_launch_us'] * w['launches_per_step'] / 1e3:.3f} ms, frac {w['frac']:.3f}"
          f"  conv frac {c['frac']:.3f}  loss {d['final_loss']:.4f}")
PY
