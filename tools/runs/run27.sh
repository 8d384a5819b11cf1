#!/bin/bash
# soak: many steps in every mode (team barriers spin in LDS: a lost wake-up would hang here, under timeout)
set -e
mkdir -p gpurun_out/r27
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --warmup 3"
timeout -k 10 120 $B --steps 600 > gpurun_out/r27/bf16.json 2>/dev/null; echo bf16 ok
timeout -k 10 120 $B --steps 400 --precision fp16 > gpurun_out/r27/fp16.json 2>/dev/null; echo fp16 ok
timeout -k 10 200 $B --steps 60 --precision fp16 --start-filts 128 --gpu-augment > gpurun_out/r27/wide.json 2>/dev/null; echo wide ok
timeout -k 10 200 $B --steps 60 --precision f32h3 > gpurun_out/r27/f32h3.json 2>/dev/null; echo f32h3 ok
timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-mode --no-infer --steps 5 --warmup 1 --tiled-pings 262144 > gpurun_out/r27/tiled.json 2>/dev/null; echo tiled ok
python - <<'PY'
import json, math
for n in ("bf16","fp16","wide","f32h3","tiled"):
    d=json.loads(open(f"gpurun_out/r27/{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), "loss", d.get("final_loss"), "finite", math.isfinite(d.get("final_loss", 0.0)), (d.get("tiled") or {}).get("patches_per_s"))
PY
