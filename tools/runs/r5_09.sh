#!/bin/bash
# fragment-major weight planes (CRIMAC_EPI_WFRAG): kernel + network tests, per-launch and timed-step A/B (CRIMAC_WFRAG=0/1)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_09; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lowp_layerwise.py -m gpu -x -q -k "fragment or halo or dgrad or cols or pack or layerwise or teacher" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log; exit 1; }
tail -2 $R/pytest.log
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q > $R/pytest_unet.log 2>&1 || { tail -30 $R/pytest_unet.log; exit 1; }
tail -2 $R/pytest_unet.log
for S in 0 1 0 1; do
  CRIMAC_WFRAG=$S timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_wfrag$S.txt 2>&1 || { tail $R/launches_wfrag$S.txt; exit 1; }
  echo "wfrag=$S $(tail -1 $R/launches_wfrag$S.txt) conv: $(grep crimac_conv3x3 $R/launches_wfrag$S.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_wfrag0.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_wfrag1.txt | awk '{print $6}')
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop"
for S in 0 1 0 1; do
  CRIMAC_WFRAG=$S timeout -k 10 200 python bench.py $BARGS > $R/bench_wfrag$S.json 2> $R/bench_wfrag$S.err || { tail $R/bench_wfrag$S.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_wfrag$S.json')); print('wfrag=$S', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4), 'conv frac', round(d['infer_patches_per_s']), 'infer', d['golden_parity']['eval_argmax_flips'], 'flips')"
done
echo r5_09 done
