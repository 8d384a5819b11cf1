#!/bin/bash
# GPU call 2: full GPU suite after the fp16 refactor, bf16 regression bench, fp16 bench, wide-net (configs[4]) bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2b
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r2b/tests.log 2>&1
rc=$?
tail -15 gpurun_out/r2b/tests.log
if [ $rc -gt 1 ]; then echo "tests killed rc=$rc"; exit $rc; fi
timeout -k 10 200 python bench.py --no-cpu-baseline --no-tiled --no-parity-mode > gpurun_out/r2b/bench_bf16.json 2> gpurun_out/r2b/bench_bf16.err || { echo bench failed; tail -20 gpurun_out/r2b/bench_bf16.err; exit 1; }
timeout -k 10 200 python bench.py --precision fp16 --no-cpu-baseline --no-tiled --no-parity-mode > gpurun_out/r2b/bench_fp16.json 2> gpurun_out/r2b/bench_fp16.err || { echo bench fp16 failed; tail -20 gpurun_out/r2b/bench_fp16.err; exit 1; }
timeout -k 10 300 python bench.py --precision fp16 --start-filts 128 --gpu-augment --no-cpu-baseline > gpurun_out/r2b/bench_wide_fp16.json 2> gpurun_out/r2b/bench_wide_fp16.err || { echo bench wide failed; tail -20 gpurun_out/r2b/bench_wide_fp16.err; exit 1; }
python - <<'PY'
import json
for n in ("bf16","fp16","wide_fp16"):
    d=json.load(open(f"gpurun_out/r2b/bench_{n}.json"))
    print(n, round(d["value"],1), "patches/s", round(d["ms_per_step"],2), "ms", "infer", round(d["infer_patches_per_s"] or 0), "conv frac", round(d["roofline"]["frac"],3), "wgrad frac", round(d["roofline_wgrad"]["frac"],3), "loss", d["final_loss"])
PY
echo run2 done
