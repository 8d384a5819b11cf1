#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2p
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r2p/tests.log 2>&1
rc=$?
tail -5 gpurun_out/r2p/tests.log
if [ $rc -gt 1 ]; then echo "tests killed rc=$rc"; exit $rc; fi
if [ $rc -ne 0 ]; then grep -n "^E \|Error\|FAILED" gpurun_out/r2p/tests.log | head -30; fi
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2p/smoke.log 2>&1; tail -2 gpurun_out/r2p/smoke.log
timeout -k 10 500 python bench.py > gpurun_out/r2p/bench.json 2> gpurun_out/r2p/bench.err || { echo bench failed; tail -20 gpurun_out/r2p/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2p/bench.json"))
print("train", round(d["value"],1), round(d["ms_per_step"],2), "infer", round(d["infer_patches_per_s"]), "tiled", round(d["tiled"]["patches_per_s"]), "parity", d["parity_mode"]["precision"], round(d["parity_mode"]["train_patches_per_s"]), round(d["parity_mode"]["infer_patches_per_s"]), "roofline", round(d["roofline"]["frac"],3), d["roofline"].get("mfma_busy_frac"), d["roofline"].get("clock_ghz"), "cpu", d["cpu_baseline"]["value"])
PY
