#!/bin/bash
# weight fragments two taps ahead (ring of three, -DCRIMAC_WCH_B3) against one tap ahead: correctness + per-launch A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_07; mkdir -p $R
CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_wchb3.so timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "halo or dgrad or cols or maxpool" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log; exit 1; }
tail -2 $R/pytest.log
for L in base b3 base b3; do
  if [ "$L" = b3 ]; then export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_wchb3.so; else unset CRIMAC_LIB; fi
  timeout -k 10 200 python tools/step_launches.py bf16 20 > $R/launches_$L.txt 2>&1 || { tail $R/launches_$L.txt; exit 1; }
  echo "variant $L: conv $(grep crimac_conv3x3 $R/launches_$L.txt | awk '{s+=$6} END {print s}') us; $(tail -1 $R/launches_$L.txt)"
done
paste <(grep crimac_conv3x3 $R/launches_base.txt | awk '{print $1, $3, $6}') <(grep crimac_conv3x3 $R/launches_b3.txt | awk '{print $6}')
echo r5_07 done
