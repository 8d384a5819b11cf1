#!/usr/bin/env python3
"""Where a workgroup of conv3x3_wch_kernel spends its cycles (diagnostic build -DCRIMAC_DIAG_CLOCK -DCRIMAC_DIAG_PHASES)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr
lib = hip.load_library()
rd = lib.crimac_diag_clock_conv_read; rd.argtypes = [C.c_void_p]; rd.restype = C.c_int
PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16"          # bf16 | h3p (plane pairs: chunks of 32 real channels)
B, P = 32, hip.PREC_NAMES[PREC]
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # 16: fragment-major planes, 48: ... and the rows form (random weights: any layout)
TILE = 512 if FLAGS & 32 else 256                             # pixels, and 64 output channels instead of 128, per workgroup
HPM = PREC == "h3p"
SHAPES = [("e1c1 64->128@128", 128, 64, 128), ("e1c2 128->128@128", 128, 128, 128), ("d2c1 256->128@128", 128, 256, 128),
          ("e2c2 256->256@64", 64, 256, 256), ("e3c2 512->512@32", 32, 512, 512), ("d0c1 1024->512@32", 32, 1024, 512)]
if HPM:
    SHAPES = [("e0c2 64->64@256", 256, 64, 64), ("d3c1 128->64@256", 256, 128, 64)] + SHAPES
for name, H, Ci, Co in SHAPES:
    M = B * H * H
    if HPM:
        v = torch.randn(M, Ci, device="cuda")
        hi = v.half(); lo = (v - hi.float()).half()
        x = torch.stack([hi.view(M, Ci // 8, 8), lo.view(M, Ci // 8, 8)], 2).contiguous().view(torch.float32).view(M, Ci)
    else:
        x = torch.randn(M, Ci, device="cuda").bfloat16()
    w = torch.randint(-3000, 3000, ((2 if HPM else 1) * 9 * Co * Ci,), dtype=torch.int16, device="cuda")
    bias = torch.randn(Co, device="cuda"); out = torch.empty(M, Co, device="cuda", dtype=torch.float32 if HPM else torch.bfloat16)
    st = torch.zeros(2, 64, Co, dtype=torch.float64, device="cuda")
    fn = lambda: call("crimac_conv3x3", P, ptr(x), Ci, B, H, H, Ci, Co, ptr(w), ptr(w), ptr(bias), ptr(out), Co, FLAGS, 1,
                      ptr(st[0]), ptr(st[1]), 64, None, 0, None, 0)
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    buf = (C.c_ulonglong * (2 * 4096))(); assert rd(buf) == 0
    nwg = min(1024, B * H * H // TILE * max(Co // (64 if FLAGS & 32 else 128), 1))
    a = np.frombuffer(buf, dtype=np.uint64)[:4096].astype(np.float64).reshape(1024, 4)[:nwg]
    m = np.median(a, axis=0); k = Ci // (32 if HPM else 64)
    w = np.frombuffer(buf, dtype=np.uint64)[4096:4096 + 2 * min(nwg, 1024)].reshape(-1, 2)
    if nwg <= 1024 and w[:, 0].min() > 0:          # (every workgroup has its own slot: entry / exit wall clock at 100 MHz)
        t0 = w[:, 0].astype(np.float64) * 0.01
        t1 = (w[:, 1] >> np.uint64(16)).astype(np.float64) * 0.01
        base = t0.min()
        st, en = np.sort(t0 - base), np.sort(t1 - base)
        print(f"{name:20s} {nwg} workgroups: entry after the first one p50 {st[len(st) // 2]:5.1f} p90 {st[int(len(st) * 0.9)]:5.1f} max {st[-1]:5.1f} us | "
              f"exit p10 {en[int(len(en) * 0.1)]:5.1f} p50 {en[len(en) // 2]:5.1f} max {en[-1]:5.1f} us | lifetime median {np.median(t1 - t0):5.1f} us", flush=True)
    print(f"{name:20s} {us:6.1f} us ({2.0 * 9 * Ci * Co * M / us / 1e6:6.0f} TFLOP/s) workgroup cycles: prologue {m[0]:6.0f} | halo waits {m[1]:6.0f} "
          f"({m[1] / k:5.0f}/chunk) | MFMA {m[2]:7.0f} ({m[2] / k:6.0f}/chunk; alone on the SIMD 9216) | epilogue {m[3]:6.0f} | sum {m.sum():7.0f}", flush=True)
