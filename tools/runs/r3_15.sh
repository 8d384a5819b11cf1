#!/bin/bash
set -e
mkdir -p gpurun_out/r3_15
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_h3p.py -q -x -m gpu > gpurun_out/r3_15/tests.log 2>&1 || { tail -60 gpurun_out/r3_15/tests.log; exit 1; }
tail -2 gpurun_out/r3_15/tests.log
CRIMAC_LIB=$PWD/gpurun_exp_diagconv.so timeout -k 10 300 python tools/diag_wch_phases.py h3p 2>&1 | grep -v amdgpu | cut -c1-250
timeout -k 10 300 python tools/check_h3p.py timing 2>&1 | grep "h3p\]"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-mode --no-tiled --no-wide --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['infer_patches_per_s'], d['roofline']['frac'])"
