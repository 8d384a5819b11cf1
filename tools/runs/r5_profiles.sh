#!/bin/bash
# round-5 profile refresh for both benched precisions: kernel stats (two streams + serialized), PMC traffic, MFMA busy
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r5_profiles
mkdir -p $R
export TMPDIR=/tmp
for P in ${PRECS:-bf16 h3f}; do
  BARGS="--precision $P --steps 4 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer --no-wide --no-train-loop --roofline-steps 3 --roofline-warmup 1"
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${P}_prof -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/${P}_prof.log 2>&1 || { echo prof failed; tail -20 $R/${P}_prof.log; exit 1; }
  CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${P}_prof_serial -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/${P}_prof_serial.log 2>&1 || { echo prof serial failed; exit 1; }
  echo "$P kernel traces ok"
  CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/${P}_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/${P}_pmc_fetch.log 2>&1 || { echo pmc fetch failed; exit 1; }
  CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/${P}_pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/${P}_pmc_write.log 2>&1 || { echo pmc write failed; exit 1; }
  CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/${P}_pmc_mfma -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/${P}_pmc_mfma.log 2>&1 || { echo pmc mfma failed; exit 1; }
  echo "$P pmc ok"
  cd $GRAFT_REPO_ROOT
  for d in prof prof_serial; do
    f=$(find $R/${P}_$d -name "*kernel_stats.csv" | head -1); cp $f $R/${P}_${d}_kernel_stats.csv
    t=$(find $R/${P}_$d -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py $t > $R/${P}_${d}_step_breakdown.txt
  done
  ff=$(find $R/${P}_pmc_fetch -name "*counter_collection.csv" | head -1)
  fw=$(find $R/${P}_pmc_write -name "*counter_collection.csv" | head -1)
  fm=$(find $R/${P}_pmc_mfma -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $ff $fw $R/${P}_pmc_traffic.json > $R/${P}_pmc_traffic.txt
  python tools/mfma_util.py $fm $R/${P}_mfma_util.json > $R/${P}_mfma_util.txt 2>&1 || { echo mfma_util failed; tail $R/${P}_mfma_util.txt; }
  cp $fm $R/${P}_pmc_mfma_busy_counter_collection.csv; cp $ff $R/${P}_pmc_fetch_counter_collection.csv; cp $fw $R/${P}_pmc_write_counter_collection.csv
  gzip -9f $R/${P}_pmc_mfma_busy_counter_collection.csv $R/${P}_pmc_fetch_counter_collection.csv $R/${P}_pmc_write_counter_collection.csv
  rm -rf $R/${P}_prof $R/${P}_prof_serial $R/${P}_pmc_fetch $R/${P}_pmc_write $R/${P}_pmc_mfma
  head -14 $R/${P}_prof_serial_step_breakdown.txt; cat $R/${P}_mfma_util.txt | tail -8; head -8 $R/${P}_pmc_traffic.txt
done
echo r5_profiles done
