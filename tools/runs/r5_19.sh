#!/bin/bash
# last decoder level's d(concat) as two launches of the persistent 64 -> 64 kernel also without a side stream: tests + per-launch table + bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_19; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_lowp_layerwise.py tests/test_gpu_nccl.py -m gpu -x -q > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log; exit 1; }
tail -2 $R/pytest.log
CRIMAC_WGRAD_STREAM=0 timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "golden or trajectory or gradient or train" > $R/pytest_serial.log 2>&1 || { tail -30 $R/pytest_serial.log; exit 1; }
tail -2 $R/pytest_serial.log
timeout -k 10 200 python tools/step_launches.py bf16 30 > $R/bf16_serial_per_launch.txt 2>&1 || { tail $R/bf16_serial_per_launch.txt; exit 1; }
grep -c crimac_conv3x3 $R/bf16_serial_per_launch.txt; echo "conv: $(grep crimac_conv3x3 $R/bf16_serial_per_launch.txt | awk '{s+=$6} END {print s}') us"; sed -n 20,30p $R/bf16_serial_per_launch.txt
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop"
for S in 1 2; do
  timeout -k 10 200 python bench.py $BARGS > $R/bench_$S.json 2> $R/bench_$S.err || { tail $R/bench_$S.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$S.json')); print(round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4), 'conv frac', d['roofline']['launches_per_step'], 'launches', round(d['roofline_wgrad']['frac'],4))"
done
echo r5_19 done
