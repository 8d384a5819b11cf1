#!/bin/bash
# round 3, run 2: seabed-mask rule, ABI check, bench line with measured golden parity
set -e
mkdir -p gpurun_out/r3_02
timeout -k 10 500 python -m pytest tests/test_tiling.py tests/test_cabi_and_host.py -q -x -m gpu > gpurun_out/r3_02/tests.log 2>&1 || { tail -30 gpurun_out/r3_02/tests.log; exit 1; }
tail -2 gpurun_out/r3_02/tests.log
timeout -k 10 500 python bench.py --steps 10 --warmup 3 > gpurun_out/r3_02/bench.json 2> gpurun_out/r3_02/bench.err || { tail -30 gpurun_out/r3_02/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_02/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["infer_patches_per_s"], d["golden_parity"])
print(d["parity_mode"]["train_patches_per_s"], d["parity_mode"]["infer_patches_per_s"], d["parity_mode"]["golden_parity"])
print(d["tiled"]["patches_per_s"], d["cpu_baseline"])
PY
