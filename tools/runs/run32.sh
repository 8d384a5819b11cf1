#!/bin/bash
set -e
mkdir -p gpurun_out/r32
timeout -k 10 300 env CRIMAC_LIB=$PWD/gpurun_exp_wgnt.so python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wgrad" 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --no-infer --steps 30"
for rep in a b; do
  timeout -k 10 200 $B > gpurun_out/r32/base$rep.json 2>/dev/null
  CRIMAC_LIB=$PWD/gpurun_exp_wgnt.so timeout -k 10 200 $B > gpurun_out/r32/nt$rep.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("basea","nta","baseb","ntb"):
    d=json.loads(open(f"gpurun_out/r32/{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline_wgrad"]["frac"],4))
PY
