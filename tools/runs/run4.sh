#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2d
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r2d/tests.log 2>&1
rc=$?
tail -12 gpurun_out/r2d/tests.log
if [ $rc -gt 1 ]; then echo "tests killed rc=$rc"; exit $rc; fi
for v in 1 0; do
CRIMAC_WGRAD_PARTIALS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-tiled --no-parity-mode > gpurun_out/r2d/bench_p$v.json 2> gpurun_out/r2d/bench_p$v.err || { echo bench failed; tail -20 gpurun_out/r2d/bench_p$v.err; exit 1; }
done
for v in 1 0; do
CRIMAC_WGRAD_PARTIALS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-tiled --no-parity-mode > gpurun_out/r2d/bench_q$v.json 2> gpurun_out/r2d/bench_q$v.err || { echo bench failed; exit 1; }
done
python - <<'PY'
import json
for n in ("p1","p0","q1","q0"):
    d=json.load(open(f"gpurun_out/r2d/bench_{n}.json"))
    print(n, round(d["value"],1), "patches/s", round(d["ms_per_step"],3), "ms", "wgrad frac", round(d["roofline_wgrad"]["frac"],3), "avg us", round(d["roofline_wgrad"]["avg_launch_us"],1), "conv", round(d["roofline"]["frac"],3))
PY
