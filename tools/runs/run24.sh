#!/bin/bash
set -e
mkdir -p gpurun_out/r24
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 20"
timeout -k 10 300 $B > gpurun_out/r24/bf16.json 2>/dev/null
timeout -k 10 300 $B --precision fp16 > gpurun_out/r24/fp16.json 2>/dev/null
timeout -k 10 400 $B --precision fp16 --start-filts 128 --gpu-augment --steps 8 > gpurun_out/r24/wide.json 2>/dev/null
timeout -k 10 300 $B --precision f32x3 --steps 8 > gpurun_out/r24/f32x3.json 2>/dev/null
timeout -k 10 300 $B --precision f32h3 --steps 8 > gpurun_out/r24/f32h3.json 2>/dev/null
python - <<'PY'
import json
for n in ("bf16","fp16","wide","f32x3","f32h3"):
    d=json.loads(open(f"gpurun_out/r24/{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), round(d.get("infer_patches_per_s",0)), round(d["roofline"]["frac"],4), round(d["roofline_wgrad"]["frac"],4), round(d["train_tflops"]), round(d.get("infer_tflops",0)))
PY
