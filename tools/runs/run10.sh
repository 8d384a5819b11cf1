#!/bin/bash
# profiles refresh for round 2 + 2-rank rehearsal of the self-spawning bench (gloo, both ranks on the one GPU)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2m
export TMPDIR=/tmp
CRIMAC_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-infer > gpurun_out/r2m/bench_2rank_gloo.json 2> gpurun_out/r2m/bench_2rank_gloo.err || { echo 2-rank rehearsal failed; tail -30 gpurun_out/r2m/bench_2rank_gloo.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r2m/bench_2rank_gloo.json'));print('2-rank rehearsal:', d['n_gpus'], d['config']['backend'], d['config']['launcher'], round(d['value'],1), 'patches/s', round(d['ms_per_step'],1),'ms')"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2m/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $GRAFT_REPO_ROOT/gpurun_out/r2m/prof.log 2>&1 || { echo prof failed; tail -20 $GRAFT_REPO_ROOT/gpurun_out/r2m/prof.log; exit 1; }
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2m/prof_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $GRAFT_REPO_ROOT/gpurun_out/r2m/prof_serial.log 2>&1 || { echo prof serial failed; exit 1; }
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2m/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $GRAFT_REPO_ROOT/gpurun_out/r2m/pmc_fetch.log 2>&1 || { echo pmc fetch failed; exit 1; }
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2m/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $GRAFT_REPO_ROOT/gpurun_out/r2m/pmc_write.log 2>&1 || { echo pmc write failed; exit 1; }
cd $GRAFT_REPO_ROOT
for d in prof prof_serial; do
  f=$(find gpurun_out/r2m/$d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r2m/${d}_kernel_stats.csv
  t=$(find gpurun_out/r2m/$d -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py $t > gpurun_out/r2m/${d}_step_breakdown.txt
  rm -f $t
done
ff=$(find gpurun_out/r2m/pmc_fetch -name "*counter_collection.csv" | head -1)
fw=$(find gpurun_out/r2m/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $ff $fw gpurun_out/r2m/pmc_traffic.json > gpurun_out/r2m/pmc_traffic.txt
gzip -9 $ff $fw
cat gpurun_out/r2m/prof_serial_step_breakdown.txt | head -30
echo run10 done
