#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2r
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_lowp_layerwise.py -m gpu -q -x > gpurun_out/r2r/tests.log 2>&1
rc=$?
tail -4 gpurun_out/r2r/tests.log
if [ $rc -ne 0 ]; then grep -n "^E \|FAILED" gpurun_out/r2r/tests.log | head -20; exit 1; fi
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-tiled --no-parity-mode > gpurun_out/r2r/bench$i.json 2> gpurun_out/r2r/bench$i.err || { echo bench failed; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r2r/bench$i.json'));print(round(d['value'],1), round(d['ms_per_step'],3), round(d['infer_patches_per_s']))"
done
