#!/bin/bash
# bottleneck launches (one workgroup of the 128-channel form per CU): the 2 x 2 form on 64-channel tiles instead (CRIMAC_CONV_DEEP_S22 =
# largest workgroup count of the 128-channel form that takes it)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_27; mkdir -p $R
CRIMAC_CONV_DEEP_S22=256 timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "golden or full_step or train_step" > $R/tests.log 2>&1 || { tail -20 $R/tests.log; exit 1; }
tail -1 $R/tests.log
for S in 0 256 512 0 256 512; do
  CRIMAC_CONV_DEEP_S22=$S timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$S.txt 2>&1 || { tail $R/launches_$S.txt; exit 1; }
  echo "deep22=$S $(tail -1 $R/launches_$S.txt) conv: $(grep crimac_conv3x3 $R/launches_$S.txt | awk '{s+=$6} END {print s}') us"
done
paste <(grep crimac_conv3x3 $R/launches_0.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_256.txt | awk '{print $6}') <(grep crimac_conv3x3 $R/launches_512.txt | awk '{print $6}')
echo r5_27 done
