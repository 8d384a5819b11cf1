#!/bin/bash
# A/B inside one call: bf16 step with the normal library vs the -DCRIMAC_EXP_NOATOMIC build of wgrad.hip (the
# weight-gradient flush skipped: results are wrong, timing is an UPPER BOUND of what hiding the flush can give).
cd "$(dirname "$0")/../.." || exit 1
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --no-wide --no-train-loop --no-infer --steps 40 --warmup 10 --roofline-steps 12 --roofline-warmup 4"
for i in 1 2 3; do
  $B > gpurun_out/r4_01_base_$i.json 2> gpurun_out/r4_01_base_$i.err || exit 1
  CRIMAC_LIB=$PWD/gpurun_exp_noatomic.so $B > gpurun_out/r4_01_noatomic_$i.json 2> gpurun_out/r4_01_noatomic_$i.err || exit 1
done
python - <<'PY'
import json, glob
for tag in ("base", "noatomic"):
    for f in sorted(glob.glob(f"gpurun_out/r4_01_{tag}_*.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        w, c = d["roofline_wgrad"], d["roofline"]
        print(tag, f"step {d['ms_per_step']:.3f} ms  wgrad serial median {w['median_launch_us']:.1f} us x{w['launches_per_step']}"
              f"  conv frac {c['frac']:.3f} wgrad frac {w['frac']:.3f} calib {c['mfma_calibration']['after']['shader_clock_ghz']:.2f} GHz")
PY
