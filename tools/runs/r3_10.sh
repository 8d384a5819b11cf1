#!/bin/bash
set -e
mkdir -p gpurun_out/r3_10
t0=$(date +%s)
timeout -k 10 700 python bench.py > gpurun_out/r3_10/bench.json 2> gpurun_out/r3_10/bench.err || { tail -30 gpurun_out/r3_10/bench.err; exit 1; }
echo "bench wall: $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_10/bench.json").read().strip().splitlines()[-1])
print("bf16", d["value"], d["infer_patches_per_s"], d["roofline"]["frac"], d["roofline_wgrad"]["frac"], d["golden_parity"]["eval_argmax_flips"])
p=d["parity_mode"]
print(p["precision"], p["train_patches_per_s"], p["infer_patches_per_s"], p["roofline"]["frac"], p["roofline_wgrad"]["frac"], p["tiled"]["patches_per_s"])
print(p["batch32_vs_oracle"])
print("tiled", d["tiled"]["patches_per_s"], "wide", d["wide_fp16"])
print(d["cpu_baseline"])
PY
