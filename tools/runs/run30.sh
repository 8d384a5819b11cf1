#!/bin/bash
set -e
mkdir -p gpurun_out/r30
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --steps 30"
for rep in a b; do
  timeout -k 10 200 $B > gpurun_out/r30/base$rep.json 2>/dev/null
  for v in p64nt wchnt; do CRIMAC_LIB=$PWD/gpurun_exp_$v.so timeout -k 10 200 $B > gpurun_out/r30/$v$rep.json 2>/dev/null; done
done
python - <<'PY'
import json
for n in ("basea","p64nta","wchnta","baseb","p64ntb","wchntb"):
    d=json.loads(open(f"gpurun_out/r30/{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), round(d["infer_patches_per_s"]))
PY
