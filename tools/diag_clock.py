#!/usr/bin/env python3
"""In-kernel clock of the two dominant MFMA kernels (MI355X_MICROARCH.md, DVFS give-back (6)).

Diagnostic builds only (tools/exp_build.sh NAME FILE -DCRIMAC_DIAG_CLOCK): conv3x3_wch_kernel / wgrad_kernel stamp
s_memtime (shader cycles) and s_memrealtime (100 MHz) around their main loop; after >= 2 s of back-to-back
launches on random data the median over workgroups of d_memtime / d_memrealtime x 100 MHz is the clock the chip
holds inside the kernel.  usage (one process per library, CRIMAC_LIB selects it):
    CRIMAC_LIB=$PWD/gpurun_exp_diagconv.so  python tools/diag_clock.py conv  out.json
    CRIMAC_LIB=$PWD/gpurun_exp_diagwgrad.so python tools/diag_clock.py wgrad out.json
A fourth argument selects the precision (bf16 default; h3p: the plane-pair form of the convolution kernel).
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr

SHAPES = {"conv": [("d0c1 1024->512@32", 32, 1024, 512), ("e3c2 512->512@32", 32, 512, 512),
                   ("e2c2 256->256@64", 64, 256, 256), ("e1c2 128->128@128", 128, 128, 128),
                   ("e0c2 64->64@256 (persistent 64-channel kernel)", 256, 64, 64)],
          "wgrad": [("d0c1 1024->512@32", 32, 1024, 512), ("e3c2 512->512@32", 32, 512, 512),
                    ("e2c2 256->256@64", 64, 256, 256), ("e0c2 64->64@256", 256, 64, 64)]}


def main():
    what, out_path = sys.argv[1], sys.argv[2]
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5
    lib = hip.load_library()
    reader = getattr(lib, f"crimac_diag_clock_{what}_read")
    reader.argtypes = [C.c_void_p]
    reader.restype = C.c_int
    prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    hp = prec == "h3p"
    if hp and what != "conv":
        raise SystemExit("h3p: the convolution kernel carries the stamps")
    B, P = 32, hip.PREC_NAMES[prec]
    res = {}
    for name, H, Ci, Co in SHAPES[what]:
        M = B * H * H
        if hp:
            if "persistent" in name:
                name = "e0c2 64->64@256 (2 x 2 wave form)"
            v = torch.randn(M, Ci, device="cuda")
            hi = v.half(); lo = (v - hi.float()).half()       # fp16 plane pairs: [8 hi][8 lo] per 8-channel group
            x = torch.stack([hi.view(M, Ci // 8, 8), lo.view(M, Ci // 8, 8)], 2).contiguous().view(torch.float32).view(M, Ci)
            w_hi = torch.randn(2 * 9 * Co * Ci, device="cuda").half().view(torch.int16)
            out = torch.empty(M, Co, device="cuda", dtype=torch.float32)
        else:
            x = torch.randn(M, Ci, device="cuda").to(torch.bfloat16)
            w_hi = torch.randn(9 * Co * Ci, device="cuda").to(torch.bfloat16).view(torch.int16)
            out = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
        dy = torch.randn(M, Co, device="cuda").to(torch.bfloat16)
        bias = torch.randn(Co, device="cuda")
        dw = torch.zeros(9 * Co * Ci, dtype=torch.float32, device="cuda")
        if what == "conv":
            fn = lambda: call("crimac_conv3x3", P, ptr(x), Ci, B, H, H, Ci, Co, ptr(w_hi), ptr(w_hi), ptr(bias),
                              ptr(out), Co, 0, 0, None, None, 64, None, 0, None, 0)
        else:
            fn = lambda: call("crimac_wgrad", P, 0, ptr(dy), Co, Co, ptr(x), Ci, Ci, B, H, H, ptr(dw), 0)
        fn()
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while time.perf_counter() - t0 < seconds:          # >= 2 s of back-to-back launches: steady-state clock
            for _ in range(50):
                fn()
            n += 50
            torch.cuda.synchronize()
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / 20
        buf = (C.c_ulonglong * (2 * 4096))()
        assert reader(buf) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 2).astype(np.float64)
        if "persistent" in name:
            a = a[:256]                # one workgroup per CU: the slots beyond hold the previous shape's stamps
        a = a[a[:, 1] > 0]
        ghz = a[:, 0] / a[:, 1] * 0.1
        flops = 2.0 * 9 * Ci * Co * M
        res[name] = {"clock_ghz_median": float(np.median(ghz)), "clock_ghz_p10": float(np.percentile(ghz, 10)),
                     "clock_ghz_p90": float(np.percentile(ghz, 90)), "workgroups_sampled": int(len(ghz)),
                     "loop_cycles_median": float(np.median(a[:, 0])), "launch_us": us,
                     "tflops": flops / us / 1e6, "launches_before_sample": n}
        # 16-bit MFMAs: 1024 flop per cycle and SIMD (a 16x16x32 MFMA = 16384 flop in 16 cycles), 1024 SIMDs -- the share of
        # the chip's MFMA issue slots the launch fills AT THE CLOCK IT RAN AT (launch overheads and tails included)
        if not hp:
            res[name]["mfma_slots_filled_at_that_clock"] = flops / us / 1e6 / (float(np.median(ghz)) * 1024 * 1024 / 1e3)
        print(f"{what:5s} {name:22s} clock {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 "
              f"{np.percentile(ghz, 90):.3f}), {us:7.1f} us, {flops / us / 1e6:7.1f} TFLOP/s"
              + ("" if hp else f", MFMA slots filled at that clock {res[name]['mfma_slots_filled_at_that_clock']:.3f}"), flush=True)
    json.dump({"what": what, "method": "s_memtime / s_memrealtime x 100 MHz around the kernel's main loop, median over "
               "workgroups, after >= 2 s of back-to-back launches on random data (diagnostic build; stamps "
               "cost wave cycles, read the clock, not the run time)", "shapes": res}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
