#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2k
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r2k/tests.log 2>&1
rc=$?
tail -8 gpurun_out/r2k/tests.log
if [ $rc -gt 1 ]; then echo "tests killed rc=$rc"; exit $rc; fi
if [ $rc -ne 0 ]; then grep -n "^E \|Error\|f32h3" gpurun_out/r2k/tests.log | head -30; fi
grep -n "eval f32h3" gpurun_out/r2k/tests.log
for pp in f32h3 f32x6 f32x3; do
timeout -k 10 300 python bench.py --precision $pp --steps 8 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode > gpurun_out/r2k/bench_$pp.json 2> gpurun_out/r2k/bench_$pp.err || { echo bench $pp failed; tail -20 gpurun_out/r2k/bench_$pp.err; exit 1; }
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 python bench.py --precision $pp --steps 8 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode > gpurun_out/r2k/bench_${pp}_serial.json 2> gpurun_out/r2k/bench_${pp}_serial.err || { echo bench $pp failed; exit 1; }
done
python - <<'PY'
import json
for n in ("f32h3","f32h3_serial","f32x6","f32x6_serial","f32x3","f32x3_serial"):
    d=json.load(open(f"gpurun_out/r2k/bench_{n}.json"))
    print(n, round(d["value"],1), "patches/s", round(d["ms_per_step"],2), "ms", "infer", round(d["infer_patches_per_s"]), "conv us", round(d["roofline"]["avg_launch_us"]), "wgrad us", round(d["roofline_wgrad"]["avg_launch_us"]))
PY
