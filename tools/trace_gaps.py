#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (one stream / serialized step):
usage: trace_gaps.py kernel_trace.csv [first_kernel_regex]   -- prints, per step, wall, busy and the gap histogram."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"nchw_to_nhwc_kernel")
starts = [i for i, r in enumerate(rows) if first.search(r["Kernel_Name"])]
def short(n):
    m = re.search(r"([a-z0-9_]+_kernel)", n)
    return m.group(1) if m else n[:28]
for a, b in list(zip(starts, starts[1:]))[-3:]:
    ks = rows[a:b]
    wall = int(ks[-1]["End_Timestamp"]) - int(ks[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
    gaps = [(int(ks[i + 1]["Start_Timestamp"]) - int(ks[i]["End_Timestamp"]), short(ks[i]["Kernel_Name"]), short(ks[i + 1]["Kernel_Name"])) for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"step: {len(ks)} kernels, wall {wall / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps>0: {len(pos)} sum {sum(g[0] for g in pos) / 1e3:.1f} us, "
          f"median {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3:.1f} us, overlaps sum {sum(-g[0] for g in gaps if g[0] < 0) / 1e3:.1f} us")
by = collections.defaultdict(list)
for g, p, n in gaps:
    by[(p, n)].append(g)
for (p, n), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"   {p:32s} -> {n:32s} x{len(v):3d}  mean gap {sum(v) / len(v) / 1e3:7.1f} us  total {sum(v) / 1e3:8.1f}")
