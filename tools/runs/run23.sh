#!/bin/bash
# round-2 final profile refresh: kernel stats (2 streams + serialized), PMC traffic, MFMA busy, in-kernel clocks
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r23
mkdir -p $R
export TMPDIR=/tmp
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/prof.log 2>&1 || { echo prof failed; tail -20 $R/prof.log; exit 1; }
echo prof ok
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_serial -- python3 $GRAFT_REPO_ROOT/bench.py $BARGS > $R/prof_serial.log 2>&1 || { echo prof serial failed; exit 1; }
echo prof serial ok
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $R/pmc_fetch.log 2>&1 || { echo pmc fetch failed; exit 1; }
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $R/pmc_write.log 2>&1 || { echo pmc write failed; exit 1; }
echo pmc traffic ok
CRIMAC_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_mfma -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tiled --no-parity-mode --no-infer > $R/pmc_mfma.log 2>&1 || { echo pmc mfma failed; exit 1; }
echo pmc mfma ok
cd $GRAFT_REPO_ROOT
for d in prof prof_serial; do
  f=$(find $R/$d -name "*kernel_stats.csv" | head -1); cp $f $R/${d}_kernel_stats.csv
  t=$(find $R/$d -name "*kernel_trace.csv" | head -1); python tools/step_breakdown.py $t > $R/${d}_step_breakdown.txt
  rm -f $t
done
ff=$(find $R/pmc_fetch -name "*counter_collection.csv" | head -1)
fw=$(find $R/pmc_write -name "*counter_collection.csv" | head -1)
fm=$(find $R/pmc_mfma -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $ff $fw $R/pmc_traffic.json > $R/pmc_traffic.txt
CRIMAC_LIB=$PWD/gpurun_exp_diagconv.so timeout -k 10 200 python tools/diag_clock.py conv $R/inkernel_clock_conv.json 2.0 > $R/clock_conv.log 2>&1 || { echo clock conv failed; tail $R/clock_conv.log; }
CRIMAC_LIB=$PWD/gpurun_exp_diagwgrad.so timeout -k 10 200 python tools/diag_clock.py wgrad $R/inkernel_clock_wgrad.json 2.0 > $R/clock_wgrad.log 2>&1 || { echo clock wgrad failed; tail $R/clock_wgrad.log; }
python tools/mfma_util.py $fm $R/mfma_util.json $R/inkernel_clock_conv.json $R/inkernel_clock_wgrad.json > $R/mfma_util.txt 2>&1 || { echo mfma_util failed; tail $R/mfma_util.txt; }
cp $fm $R/pmc_mfma_busy_counter_collection.csv; cp $ff $R/pmc_fetch_counter_collection.csv; cp $fw $R/pmc_write_counter_collection.csv
gzip -9f $R/pmc_mfma_busy_counter_collection.csv $R/pmc_fetch_counter_collection.csv $R/pmc_write_counter_collection.csv
rm -rf $R/prof $R/prof_serial $R/pmc_fetch $R/pmc_write $R/pmc_mfma
head -12 $R/prof_serial_step_breakdown.txt; cat $R/mfma_util.txt | tail -12; grep -v amdgpu $R/clock_conv.log | tail -6; grep -v amdgpu $R/clock_wgrad.log | tail -5
echo run23 done
