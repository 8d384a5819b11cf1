#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2g
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r2g/tests.log 2>&1
rc=$?
tail -12 gpurun_out/r2g/tests.log
if [ $rc -gt 1 ]; then echo "tests killed rc=$rc"; exit $rc; fi
if [ $rc -ne 0 ]; then grep -n "^E \|Error" gpurun_out/r2g/tests.log | head -30; fi
timeout -k 10 400 python bench.py > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err || { echo bench failed; tail -20 gpurun_out/r2g/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2g/bench.json"))
print("train", round(d["value"],1), "infer", round(d["infer_patches_per_s"]), "tiled", d["tiled"]["patches_per_s"], "parity", d["parity_mode"]["train_patches_per_s"], d["parity_mode"]["infer_patches_per_s"])
PY
