#!/bin/bash
set -e
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --no-infer --steps 30"
CRIMAC_WGRAD_TEAMS=1 timeout -k 10 200 $B > gpurun_out/r19_t1.json
timeout -k 10 200 $B > gpurun_out/r19_t2.json
CRIMAC_WGRAD_PARTIALS=1 timeout -k 10 200 $B > gpurun_out/r19_t2p.json
CRIMAC_WGRAD_TEAMS=1 timeout -k 10 200 $B > gpurun_out/r19_t1b.json
timeout -k 10 200 $B > gpurun_out/r19_t2b.json
CRIMAC_WGRAD_PARTIALS=1 timeout -k 10 200 $B > gpurun_out/r19_t2pb.json
python - <<'PY'
import json
for n in ("t1","t2","t2p","t1b","t2b","t2pb"):
    d=json.loads(open(f"gpurun_out/r19_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), d["roofline_wgrad"]["frac"], d["roofline_wgrad"]["avg_launch_us"])
PY
