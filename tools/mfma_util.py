#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass.

usage: mfma_util.py counter_collection.csv out.json [inkernel_conv.json inkernel_wgrad.json]

  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
      (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs and the SQ counter over all SIMDs; the busy counter counts
       cycles: 16 per v_mfma_f32_16x16x32_bf16, MI355X_MICROARCH.md cycle constants)
  clock_ghz (PMC)  = GRBM_GUI_ACTIVE / 8 / dispatch time -- reads HIGH on dispatches shorter than ~0.3 ms (guide,
      DVFS give-back); the in-kernel s_memtime / s_memrealtime figure (tools/diag_clock.py) is the one to trust and
      is merged in when given.
"""
import collections
import csv
import json
import re
import sys

GROUPS = {"conv3x3": r"conv3x3_(wch|p64|glds_w4|c16)_kernel|conv3x3_kernel", "conv3x3_wch": r"conv3x3_wch_kernel",
          "conv3x3_p64": r"conv3x3_p64_kernel", "conv3x3_w4": r"conv3x3_glds_w4_kernel",
          "wgrad": r"wgrad(_up_pp|_pp)?_kernel", "wgrad_group": r"wgrad(_pp)?_group_kernel", "wgrad_pp": r"wgrad_pp_kernel", "upconv": r"upconv_wch_kernel"}


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in rows:
        for g, pat in GROUPS.items():
            if re.search(pat, r["Kernel_Name"]):
                per[g][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen[g]:
                    seen[g].add(r["Dispatch_Id"])
                    per[g]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    per[g]["launches"] += 1
    out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on CRIMAC_WGRAD_STREAM=0 python3 bench.py "
                   "(serialized); mfma_busy_frac = busy cycles / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); clock_ghz = in-kernel "
                   "s_memtime/s_memrealtime where measured (tools/diag_clock.py), else GRBM_GUI_ACTIVE/8/time (reads high "
                   "on sub-0.3-ms dispatches)"}
    for g, v in per.items():
        busy, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        out[g] = {"launches": int(v["launches"]), "avg_launch_us": v["ns"] / v["launches"] / 1e3,
                  "mfma_busy_cycles": busy, "grbm_gui_active": gui,
                  "mfma_busy_frac": busy / (gui / 8.0 * 1024.0),
                  "clock_ghz_pmc": gui / 8.0 / v["ns"], "clock_ghz": gui / 8.0 / v["ns"],
                  "mfma_equiv_tflops_at_busy": busy * 1024.0 / (v["ns"] * 1e-9) / 1e12 if v["ns"] else None}
        print(f"{g:12s} launches {int(v['launches']):4d} avg {v['ns'] / v['launches'] / 1e3:8.1f} us  mfma_busy "
              f"{out[g]['mfma_busy_frac']:.3f}  clock(pmc) {out[g]['clock_ghz_pmc']:.2f} GHz")
    for path, keys in zip(sys.argv[3:5], (("conv3x3", "conv3x3_wch"), ("wgrad",))):
        try:
            d = json.load(open(path))["shapes"]
            ghz = sorted(s["clock_ghz_median"] for s in d.values())
            for k in keys:
                if k in out:
                    out[k]["clock_ghz"] = ghz[len(ghz) // 2]
                    # the same busy cycles against the cycles the chip really ran (in-kernel clock x dispatch time)
                    out[k]["mfma_busy_frac_at_inkernel_clock"] = out[k]["mfma_busy_cycles"] / (
                        out[k]["avg_launch_us"] * 1e-6 * out[k]["launches"] * out[k]["clock_ghz"] * 1e9 * 1024.0)
                    out[k]["clock_ghz_inkernel_by_shape"] = {n: s["clock_ghz_median"] for n, s in d.items()}
        except Exception as e:  # noqa: BLE001
            print("no in-kernel clock from", path, e)
    json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
