#!/bin/bash
# rows form of the channel-split convolution (CRIMAC_CONV_ROWS=1: every legal layer): whole-net tests, per-launch and step A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_23; mkdir -p $R
CRIMAC_CONV_ROWS=1 timeout -k 10 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_lowp_layerwise.py -m gpu -x -q --deselect "tests/test_gpu_unet.py::test_row_major_fallback_of_the_weight_planes_is_bit_identical" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log | cut -c1-250; exit 1; }
tail -2 $R/pytest.log
for S in "" 1 "" 1; do
  CRIMAC_CONV_ROWS=$S timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_r$S.txt 2>&1 || { tail $R/launches_r$S.txt; exit 1; }
  echo "rows=$S $(tail -1 $R/launches_r$S.txt) conv: $(grep crimac_conv3x3 $R/launches_r$S.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_r.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_r1.txt | awk '{print $6}')
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop"
for S in "" 1 "" 1; do
  CRIMAC_CONV_ROWS=$S timeout -k 10 200 python bench.py $BARGS > $R/bench_r$S.json 2> $R/bench_r$S.err || { tail $R/bench_r$S.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_r$S.json')); print('rows=$S', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4), 'conv frac', round(d['infer_patches_per_s']), 'infer')"
done
echo r5_23 done
