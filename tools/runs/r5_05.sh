#!/bin/bash
# short form (FORM 3, three workgroups per CU) of the channel-split convolution: kernel tests, then A/B of the serialized
# per-launch table and of the timed step with CRIMAC_CONV_SHORT = 0 / 1 / 2
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_05; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "halo or dgrad or cols or maxpool" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log; exit 1; }
tail -3 $R/pytest.log
for S in 0 2 1 0 2; do
  CRIMAC_CONV_SHORT=$S timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_short$S.txt 2>&1 || { tail $R/launches_short$S.txt; exit 1; }
  echo "short=$S $(tail -1 $R/launches_short$S.txt) conv: $(grep crimac_conv3x3 $R/launches_short$S.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep -n crimac_conv3x3 $R/launches_short0.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_short2.txt | awk '{print $6}') <(grep crimac_conv3x3 $R/launches_short1.txt | awk '{print $6}')
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop"
for S in 0 2 0 2; do
  CRIMAC_CONV_SHORT=$S timeout -k 10 200 python bench.py $BARGS > $R/bench_short$S.json 2> $R/bench_short$S.err || { tail $R/bench_short$S.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_short$S.json')); print('short=$S', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4), 'conv frac', round(d['infer_patches_per_s']), 'infer')"
done
echo r5_05 done
