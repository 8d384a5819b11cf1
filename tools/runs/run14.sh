#!/bin/bash
# fused eval max-pool: parity tests, then inference A/B
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fused_maxpool or conv3x3" > gpurun_out/r14_tests.log 2>&1 || { tail -30 gpurun_out/r14_tests.log; exit 1; }
tail -2 gpurun_out/r14_tests.log
timeout -k 10 400 python -m pytest tests/test_gpu_unet.py -q -x -m gpu -k "predict or infer or eval or golden" > gpurun_out/r14_unet.log 2>&1 || { tail -30 gpurun_out/r14_unet.log; exit 1; }
tail -2 gpurun_out/r14_unet.log
CRIMAC_FUSE_EVAL_POOL=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30 > gpurun_out/r14_bench_off.json
timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30 > gpurun_out/r14_bench_on.json
python - <<'PY'
import json
for n in ("off","on"):
    d=json.loads(open(f"gpurun_out/r14_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d.get("infer_patches_per_s"))
PY
