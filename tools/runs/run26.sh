#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r26; mkdir -p $R
export TMPDIR=/tmp
cd /tmp
CRIMAC_EVAL_SPLIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -- python3 $GRAFT_REPO_ROOT/tools/profile_infer.py > $R/prof.log 2>&1 || { echo failed; tail $R/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/infer_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r26/infer_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} total {float(r["TotalDurationNs"])/1e3/6:9.1f} us/fwd  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
print("sum per forward", tot/1e3/6, "us")
PY
rm -rf $R/prof
