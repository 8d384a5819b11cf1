#!/bin/bash
# streaming (non-temporal) loads in the elementwise kernels: tests, then whole-step / inference A/B against the plain form
set -e
mkdir -p gpurun_out/r29
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "pack or unpack or wgrad" 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30"
for rep in a b; do
  CRIMAC_LIB=$PWD/gpurun_exp_nontp.so timeout -k 10 200 $B > gpurun_out/r29/plain$rep.json 2>/dev/null
  timeout -k 10 200 $B > gpurun_out/r29/nt$rep.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("plaina","nta","plainb","ntb"):
    d=json.loads(open(f"gpurun_out/r29/{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), round(d["infer_patches_per_s"]))
PY
