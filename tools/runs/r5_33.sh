#!/bin/bash
# collate_float32 (float64 crops cast per sample in the workers' collate): the raw-crop pipeline tests and the train_loop legs
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_33; mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "raw_crops or train_model or stager or ring" > $R/pytest.log 2>&1 || { tail -30 $R/pytest.log | cut -c1-250; exit 1; }
tail -2 $R/pytest.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tiled --no-parity-mode --no-wide --no-infer > $R/bench.json 2> $R/bench.err || { tail -20 $R/bench.err; exit 1; }
grep "timed region\|train_loop" $R/bench.err
echo r5_33 done
