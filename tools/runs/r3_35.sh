#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r3_35
mkdir -p $R
export TMPDIR=/tmp
cd /tmp
CRIMAC_EVAL_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/p -- python3 $GRAFT_REPO_ROOT/tools/profile_infer.py h3p > $R/p.log 2>&1 || { echo prof failed; tail -20 $R/p.log; exit 1; }
t=$(find $R/p -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
# last forward: from the last nchw_to_nhwc to the end
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name']]
step = rows[idx[-1]:]
tot = 0
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += d
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70]
    print(f"{n:72s} grid {r['Grid_Size_X']:>9s}x{r['Grid_Size_Y']:>3s} {d:8.1f} us")
print("kernels", len(step), "sum", tot, "wall", (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3)
PY
rm -rf $R/p
