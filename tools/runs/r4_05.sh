#!/bin/bash
# kernel-to-kernel idle time of the serialized bf16 / h3f step (rocprofv3 kernel trace)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r4_05; mkdir -p $R; export TMPDIR=/tmp
for P in ${PRECS:-bf16}; do
  BARGS="--precision $P --steps 4 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide --no-train-loop --roofline-steps 3 --roofline-warmup 1"
  cd /tmp
  CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/${P}_t -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/${P}_t.log 2>&1 || { echo trace failed; exit 1; }
  cd $GRAFT_REPO_ROOT
  t=$(find $R/${P}_t -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $t > $R/${P}_gaps.txt; cat $R/${P}_gaps.txt
  rm -rf $R/${P}_t
done
