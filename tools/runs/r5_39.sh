#!/bin/bash
# persistent 64 -> 64 kernel with the weight fragment as the MFMA's A operand (packed staging writes): kernel + network tests,
# per-launch A/B against the previous kernel (gpurun_exp_p64old.so), step and inference A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_39; mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_lowp_layerwise.py -m gpu -x -q > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log | cut -c1-250; exit 1; }
tail -2 $R/pytest.log
for V in old new old new; do
  if [ "$V" = new ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_p64old.so; fi
  timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$V.txt 2>&1 || { tail $R/launches_$V.txt; exit 1; }
  echo "$V $(tail -1 $R/launches_$V.txt) conv: $(grep crimac_conv3x3 $R/launches_$V.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_old.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_new.txt | awk '{print $6}') | awk '($4 < 0.97 * $3 || $4 > 1.03 * $3)'
BARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop"
for V in old new old new; do
  if [ "$V" = new ]; then unset CRIMAC_LIB; else export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_p64old.so; fi
  timeout -k 10 200 python bench.py $BARGS > $R/bench_$V.json 2> $R/bench_$V.err || { tail $R/bench_$V.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$V.json')); print('$V', round(d['ms_per_step'],3), 'ms', round(d['infer_patches_per_s']), 'infer', round(d['roofline']['frac'],4))"
done
echo r5_39 done
