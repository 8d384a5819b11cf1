#!/bin/bash
# whole-step A/B: new library vs the previous persistent 64-channel kernel
set -e
mkdir -p gpurun_out
CRIMAC_LIB=$PWD/gpurun_exp_p64old.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30 > gpurun_out/r17_bench_old.json
timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30 > gpurun_out/r17_bench_new.json
CRIMAC_LIB=$PWD/gpurun_exp_p64old.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30 > gpurun_out/r17_bench_old2.json
timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30 > gpurun_out/r17_bench_new2.json
python - <<'PY'
import json
for n in ("old","new","old2","new2"):
    d=json.loads(open(f"gpurun_out/r17_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), round(d.get("infer_patches_per_s",0)), d["roofline"]["frac"])
PY
