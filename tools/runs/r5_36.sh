#!/bin/bash
# transposed convolutions: three A-chunk buffers, chunk c + 2 requested at the top of chunk c (CRIMAC_UPCONV_DEEP = smallest
# chunk count that takes the form): kernel + network tests with every launch on it, per-launch A/B for thresholds
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_36; mkdir -p $R
CRIMAC_UPCONV_DEEP=1 timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_lowp_layerwise.py -m gpu -x -q -k "upconv or up_conv or transposed or golden or full_step or train_step or layer" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log | cut -c1-250; exit 1; }
tail -2 $R/pytest.log
for S in 0 8 4 2 0 8 4 2; do
  CRIMAC_UPCONV_DEEP=$S timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$S.txt 2>&1 || { tail $R/launches_$S.txt; exit 1; }
  echo "deep=$S $(tail -1 $R/launches_$S.txt)"
done
paste <(grep "igemm\|upconv" $R/launches_0.txt | awk '{print $1, $2, $6}') <(grep "igemm\|upconv" $R/launches_8.txt | awk '{print $6}') <(grep "igemm\|upconv" $R/launches_4.txt | awk '{print $6}') <(grep "igemm\|upconv" $R/launches_2.txt | awk '{print $6}')
echo r5_36 done
