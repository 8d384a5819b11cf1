#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2q
export TMPDIR=/tmp
cd /tmp
CRIMAC_EVAL_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2q/prof_infer -- python3 $GRAFT_REPO_ROOT/tools/profile_infer.py > $GRAFT_REPO_ROOT/gpurun_out/r2q/prof_infer.log 2>&1 || { echo failed; tail $GRAFT_REPO_ROOT/gpurun_out/r2q/prof_infer.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r2q/prof_infer -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
find gpurun_out/r2q/prof_infer -name "*kernel_trace.csv" -delete
