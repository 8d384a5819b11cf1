#!/bin/bash
# full GPU suite + driver-style bench
set -e
mkdir -p gpurun_out/r3_07
timeout -k 10 1000 python -m pytest tests -q -x -m gpu > gpurun_out/r3_07/tests.log 2>&1 || { tail -60 gpurun_out/r3_07/tests.log; exit 1; }
tail -3 gpurun_out/r3_07/tests.log
timeout -k 10 500 python bench.py > gpurun_out/r3_07/bench.json 2> gpurun_out/r3_07/bench.err || { tail -30 gpurun_out/r3_07/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_07/bench.json").read().strip().splitlines()[-1])
print("bf16", d["value"], d["infer_patches_per_s"], d["roofline"]["frac"], d["roofline_wgrad"]["frac"])
p=d["parity_mode"]
print(p["precision"], p["train_patches_per_s"], p["infer_patches_per_s"], p["roofline"]["frac"], p["roofline_wgrad"]["frac"], p["golden_parity"])
print("tiled", d["tiled"]["patches_per_s"])
PY
