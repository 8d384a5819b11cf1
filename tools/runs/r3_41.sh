#!/bin/bash
set -e
mkdir -p gpurun_out/r3_41
timeout -k 10 900 python -m pytest tests/test_gpu_h3p.py -q -x -m gpu > gpurun_out/r3_41/t1.log 2>&1 || { tail -60 gpurun_out/r3_41/t1.log; exit 1; }
tail -1 gpurun_out/r3_41/t1.log
BA="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide"
for i in 1 2 3; do
for V in 1 0; do
CRIMAC_WGRAD_UP_PP=$V timeout -k 10 200 python bench.py --precision h3p $BA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h3p up_pp=$V', d['value'], d['ms_per_step'], d['roofline_wgrad']['frac'])"
done; done
bash tools/runs/r3_23.sh h3p | grep -E "wgrad_up|wgrad_kernel|step wall"
