#!/bin/bash
# full GPU suite + smoke + whole-step A/B vs the library of the round's start
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r21
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r21/tests.log 2>&1
rc=$?
tail -3 gpurun_out/r21/tests.log
if [ $rc -gt 1 ]; then echo "tests killed rc=$rc"; exit $rc; fi
if [ $rc -ne 0 ]; then grep -n "^E \|Error\|FAILED" gpurun_out/r21/tests.log | head -30; fi
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r21/smoke.log 2>&1; tail -1 gpurun_out/r21/smoke.log
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30"
CRIMAC_LIB=$PWD/gpurun_exp_p64old.so timeout -k 10 200 $B > gpurun_out/r21/old.json 2>/dev/null
timeout -k 10 200 $B > gpurun_out/r21/new.json 2>/dev/null
CRIMAC_LIB=$PWD/gpurun_exp_p64old.so timeout -k 10 200 $B > gpurun_out/r21/old2.json 2>/dev/null
timeout -k 10 200 $B > gpurun_out/r21/new2.json 2>/dev/null
python - <<'PY'
import json
for n in ("old","new","old2","new2"):
    d=json.loads(open(f"gpurun_out/r21/{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), round(d.get("infer_patches_per_s",0)), round(d["roofline"]["frac"],4), round(d["roofline_wgrad"]["frac"],4))
PY
