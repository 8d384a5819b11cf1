#!/bin/bash
# A/B inside one call: (1) the channel-split convolution's tile loop (CRIMAC_WCH_LOOP), bf16 and h3p, train + inference;
# (2) grouped weight gradients with bounded items per workgroup (CRIMAC_WGRAD_GROUP_WGITEMS), bf16 and h3p.
cd "$(dirname "$0")/../.." || exit 1
run() {   # tag, precision, env...
  tag=$1; prec=$2; shift 2
  env "$@" python bench.py --precision $prec --no-cpu-baseline --no-parity-mode --no-tiled --no-wide --no-train-loop \
      --steps 30 --warmup 8 --roofline-steps 10 --roofline-warmup 3 > gpurun_out/r4_04_$tag.json 2> gpurun_out/r4_04_$tag.err \
      || { echo "$tag failed"; tail -5 gpurun_out/r4_04_$tag.err; exit 1; }
}
for i in 1 2; do
  run bf16_base_$i bf16 CRIMAC_WCH_LOOP=0 CRIMAC_WGRAD_GROUP_WGITEMS=0
  run bf16_loop_$i bf16 CRIMAC_WCH_LOOP=1 CRIMAC_WGRAD_GROUP_WGITEMS=0
  run bf16_wg3_$i bf16 CRIMAC_WCH_LOOP=0 CRIMAC_WGRAD_GROUP_WGITEMS=3
  run bf16_wg2_$i bf16 CRIMAC_WCH_LOOP=0 CRIMAC_WGRAD_GROUP_WGITEMS=2
  run bf16_wg1_$i bf16 CRIMAC_WCH_LOOP=0 CRIMAC_WGRAD_GROUP_WGITEMS=1
  run bf16_both_$i bf16 CRIMAC_WCH_LOOP=1 CRIMAC_WGRAD_GROUP_WGITEMS=3
  run h3p_off_$i h3p CRIMAC_WCH_LOOP=0 CRIMAC_WGRAD_GROUP=0
  run h3p_loop_$i h3p CRIMAC_WCH_LOOP=1 CRIMAC_WGRAD_GROUP=0
  run h3p_wg3_$i h3p CRIMAC_WCH_LOOP=0 CRIMAC_WGRAD_GROUP=1 CRIMAC_WGRAD_GROUP_WGITEMS=3 CRIMAC_WGRAD_GROUP_ITEMS=256
  run h3p_wg1_$i h3p CRIMAC_WCH_LOOP=0 CRIMAC_WGRAD_GROUP=1 CRIMAC_WGRAD_GROUP_WGITEMS=1 CRIMAC_WGRAD_GROUP_ITEMS=256
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r4_04_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    w, c = d["roofline_wgrad"], d["roofline"]
    print(f"{os.path.basename(f)[6:-5]:14s} step {d['ms_per_step']:.3f} ms infer {d['infer_patches_per_s']:.0f}  conv: serial sum {c['median_launch_us'] * c['launches_per_step'] / 1e3:.3f} ms frac {c['frac']:.3f}"
          f"  wgrad: {w['launches_per_step']} launches, serial sum {w['median_launch_us'] * w['launches_per_step'] / 1e3:.3f} ms, frac {w['frac']:.3f}  loss {d['final_loss']:.4f}")
PY
