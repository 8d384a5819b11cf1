#!/bin/bash
# GPU call 1 of round 2: full GPU test suite, default bench, in-kernel clock diagnostics, MFMA-busy PMC pass.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/tests.log 2>&1
rc=$?
tail -5 gpurun_out/r2a/tests.log
if [ $rc -gt 1 ]; then echo "tests killed rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err || { echo bench failed; tail -20 gpurun_out/r2a/bench.err; exit 1; }
tail -3 gpurun_out/r2a/bench.err
CRIMAC_LIB=$PWD/gpurun_exp_diagconv.so timeout -k 10 120 python tools/diag_clock.py conv gpurun_out/r2a/clock_conv.json > gpurun_out/r2a/clock_conv.log 2>&1 || { echo diag conv failed; tail gpurun_out/r2a/clock_conv.log; exit 1; }
CRIMAC_LIB=$PWD/gpurun_exp_diagwgrad.so timeout -k 10 120 python tools/diag_clock.py wgrad gpurun_out/r2a/clock_wgrad.json > gpurun_out/r2a/clock_wgrad.log 2>&1 || { echo diag wgrad failed; tail gpurun_out/r2a/clock_wgrad.log; exit 1; }
cat gpurun_out/r2a/clock_conv.log gpurun_out/r2a/clock_wgrad.log
cd /tmp
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_mfma -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_mfma.log 2>&1 || { echo pmc failed; tail -20 $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_mfma.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r2a/pmc_mfma -name "*counter_collection.csv" | head -1)
python tools/mfma_util.py $f gpurun_out/r2a/mfma_util.json gpurun_out/r2a/clock_conv.json gpurun_out/r2a/clock_wgrad.json
gzip -9 $f
echo run1 done
