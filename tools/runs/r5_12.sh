#!/bin/bash
# unpool_add with packed y vectors (16-bit storage, fused sums): tests, micro-benchmark, step A/B against the previous library
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_12; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lowp_layerwise.py -m gpu -x -q -k "unpool or batchnorm or layerwise or teacher" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log; exit 1; }
tail -2 $R/pytest.log
timeout -k 10 200 python tools/bench_unpool.py bf16 > $R/unpool_new.txt 2>&1 || { tail $R/unpool_new.txt; exit 1; }
CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_prev.so timeout -k 10 200 python tools/bench_unpool.py bf16 > $R/unpool_old.txt 2>&1 || { tail $R/unpool_old.txt; exit 1; }
echo "--- new"; cat $R/unpool_new.txt; echo "--- old"; cat $R/unpool_old.txt
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-train-loop --no-infer"
for L in old new old new; do
  if [ "$L" = old ]; then export CRIMAC_LIB=$GRAFT_REPO_ROOT/gpurun_exp_prev.so; else unset CRIMAC_LIB; fi
  timeout -k 10 200 python bench.py $BARGS > $R/bench_$L.json 2> $R/bench_$L.err || { tail $R/bench_$L.err; exit 1; }
  python -c "
import json; d=json.load(open('$R/bench_$L.json')); print('$L', round(d['ms_per_step'],3), 'ms')"
done
echo r5_12 done
