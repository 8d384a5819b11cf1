#!/bin/bash
# fragment-major planes extended to plane pairs (h3p / h3f forward), the h3f backward and the eval packs: tests + A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_10; mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_gpu_h3p.py tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_tiling.py tests/test_lmi.py -m gpu -x -q > $R/pytest.log 2>&1 || { tail -40 $R/pytest.log; exit 1; }
tail -2 $R/pytest.log
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop"
for P in h3f bf16; do
for S in 0 1 0 1; do
  CRIMAC_WFRAG=$S timeout -k 10 300 python bench.py --precision $P $BARGS > $R/bench_${P}_wfrag$S.json 2> $R/bench_${P}_wfrag$S.err || { tail $R/bench_${P}_wfrag$S.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_${P}_wfrag$S.json')); print('$P wfrag=$S', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4), 'conv frac', round(d['infer_patches_per_s']), 'infer', d['golden_parity']['eval_argmax_flips'], 'flips', d['golden_parity']['eval_logits_rel'])"
done
done
echo r5_10 done
