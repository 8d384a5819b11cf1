#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

FETCH_SIZE / WRITE_SIZE are in KiB.  Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: on
gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane,
global_load and LDS-DMA alike) -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores and float
atomics.  Writes profiles/<tag>_pmc_traffic.json: bytes per launch, averaged over the launches.
usage: pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json
"""
import collections, csv, json, re, sys

def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(conv3x3_wch_kernel|conv3x3_glds_w4_kernel|conv3x3_c16_kernel|conv3x3_p64_kernel|conv3x3_kernel|upconv_wch_kernel|wgrad_pp_group_kernel|wgrad_group_kernel|wgrad_up_pp_kernel|wgrad_pp_kernel|wgrad_kernel|igemm_kernel|unpool_bn_bwd_apply_kernel|bn_\w+_kernel|colstats_kernel|"
                      r"unpool_add_kernel|head_\w+_kernel|sgd_kernel|pack_\w+_kernel|unpack_\w+_kernel)", r["Kernel_Name"])
        if not m:
            continue
        tot[m.group(1)] += float(r["Counter_Value"])
        n[m.group(1)] += 1
    return tot, n

fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    rd = 2.0 * fetch[k] * 1024 / nf[k]            # gfx950 correction: x2
    wr = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
    out[k] = {"launches": nf[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": rd + wr}
    print(f"{k:28s} launches {nf[k]:5d}  read {rd / 1e6:9.1f} MB  write {wr / 1e6:9.1f} MB per launch")
json.dump({"note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported; bytes per launch "
                   "averaged over all launches of the kernel in bench.py --steps 3 --warmup 1",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
