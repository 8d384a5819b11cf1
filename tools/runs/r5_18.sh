#!/bin/bash
# weights of the channel-split convolution through a private LDS ring (CRIMAC_WCH_WL=1): correctness, per-launch and step A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_18; mkdir -p $R
CRIMAC_WCH_WL=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fragment or halo or dgrad or cols or maxpool" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log; exit 1; }
tail -2 $R/pytest.log
CRIMAC_WCH_WL=1 timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_lowp_layerwise.py -m gpu -x -q > $R/pytest2.log 2>&1 || { tail -30 $R/pytest2.log; exit 1; }
tail -2 $R/pytest2.log
for S in 0 1 0 1; do
  CRIMAC_WCH_WL=$S timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_wl$S.txt 2>&1 || { tail $R/launches_wl$S.txt; exit 1; }
  echo "wl=$S $(tail -1 $R/launches_wl$S.txt) conv: $(grep crimac_conv3x3 $R/launches_wl$S.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_wl0.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_wl1.txt | awk '{print $6}')
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop"
for S in 0 1 0 1; do
  CRIMAC_WCH_WL=$S timeout -k 10 200 python bench.py $BARGS > $R/bench_wl$S.json 2> $R/bench_wl$S.err || { tail $R/bench_wl$S.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_wl$S.json')); print('wl=$S', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4), 'conv frac', round(d['infer_patches_per_s']), 'infer')"
done
echo r5_18 done
