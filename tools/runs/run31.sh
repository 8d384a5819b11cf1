#!/bin/bash
set -e
mkdir -p gpurun_out/r31
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --no-infer --steps 30"
for rep in a b; do
  for v in 512 1024 768; do CRIMAC_WGRAD_BLOCKS=$v timeout -k 10 200 $B > gpurun_out/r31/b$v$rep.json 2>/dev/null; done
done
python - <<'PY'
import json
for rep in "ab":
    for v in (512,1024,768):
        d=json.loads(open(f"gpurun_out/r31/b{v}{rep}.json").read().strip().splitlines()[-1])
        print(v, rep, round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline_wgrad"]["frac"],4))
PY
