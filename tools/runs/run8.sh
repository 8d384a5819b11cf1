#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2h
timeout -k 10 600 python -m pytest tests/test_tiling.py tests/test_gpu_unet.py -m gpu -q -x -k "tiling or chunk or memm or survey or configs3 or gather or rank or evaluate" > gpurun_out/r2h/tests.log 2>&1
rc=$?
tail -6 gpurun_out/r2h/tests.log
if [ $rc -gt 1 ]; then echo "tests killed rc=$rc"; exit $rc; fi
if [ $rc -ne 0 ]; then grep -n "^E \|Error" gpurun_out/r2h/tests.log | head -30; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-mode > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err || { echo bench failed; tail -20 gpurun_out/r2h/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2h/bench.json"))
print("train", round(d["value"],1), "infer", round(d["infer_patches_per_s"]), "tiled", d["tiled"])
PY
