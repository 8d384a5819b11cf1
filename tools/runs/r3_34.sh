#!/bin/bash
set -e
mkdir -p gpurun_out/r3_34
CRIMAC_WGRAD_PARTIALS=1 timeout -k 10 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_h3p.py -q -x -m gpu > gpurun_out/r3_34/t1.log 2>&1 || { tail -60 gpurun_out/r3_34/t1.log; exit 1; }
tail -1 gpurun_out/r3_34/t1.log
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wgrad or partial or reproduc" > gpurun_out/r3_34/t2.log 2>&1 || { tail -60 gpurun_out/r3_34/t2.log; exit 1; }
tail -1 gpurun_out/r3_34/t2.log
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for i in 1 2; do
for P in bf16 h3p; do
for W in 0 1; do
CRIMAC_WGRAD_PARTIALS=$W timeout -k 10 200 python bench.py --precision $P $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P partials=$W', d['value'], d['ms_per_step'], d['roofline_wgrad']['frac'])"
done; done; done
bash tools/runs/r3_32.sh | grep -E "step wall|wgrad|unpack|fold"
