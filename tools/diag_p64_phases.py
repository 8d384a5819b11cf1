#!/usr/bin/env python3
"""Where a tile of the persistent 64->64 kernel spends its cycles (diagnostic build -DCRIMAC_DIAG_CLOCK
-DCRIMAC_DIAG_PHASES, CRIMAC_LIB selects it): s_memtime at the phase boundaries, wave 0 of each team."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr
lib = hip.load_library()
rd = lib.crimac_diag_clock_conv_read; rd.argtypes = [C.c_void_p]; rd.restype = C.c_int
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P = hip.PREC_NAMES["bf16"]; H = 256; Cc = 64; M = B * H * H
x = torch.randn(M, Cc, device="cuda").bfloat16(); y = torch.randn(M, Cc, device="cuda").bfloat16()
w = torch.randint(-3000, 3000, (9 * Cc * Cc,), dtype=torch.int16, device="cuda")
bias = torch.randn(Cc, device="cuda"); out = torch.empty(M, Cc, device="cuda", dtype=torch.bfloat16)
st = torch.zeros(2, 64, Cc, dtype=torch.float64, device="cuda"); vec = torch.rand(4, Cc, device="cuda") + 0.5
forms = {"plain": lambda: call("crimac_conv3x3", P, ptr(x), Cc, B, H, H, Cc, Cc, ptr(w), ptr(w), ptr(bias), ptr(out), Cc, 1, 0, None, None, 64, None, 0, None, 0),
         "stats": lambda: call("crimac_conv3x3", P, ptr(x), Cc, B, H, H, Cc, Cc, ptr(w), ptr(w), ptr(bias), ptr(out), Cc, 0, 1, ptr(st[0]), ptr(st[1]), 64, None, 0, None, 0),
         "bnb": lambda: call("crimac_conv3x3", P, ptr(x), Cc, B, H, H, Cc, Cc, ptr(w), ptr(w), None, ptr(out), Cc, 0, 2, ptr(st[0]), ptr(st[1]), 64, ptr(y), Cc, ptr(vec), Cc)}
for name, fn in forms.items():
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2 * 4096))(); assert rd(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 2).astype(np.float64)[:2560].reshape(512, 5, 2)
    per = a[:, :, 0] / a[:, :, 1]
    med = np.median(per, axis=0)
    raw = np.frombuffer(buf, dtype=np.uint64)[5120:5120 + 2048].astype(np.int64).reshape(256, 2, 4)
    t0 = raw[:, :, 0].min()
    r = (raw - t0) / 100.0                                    # us since the first workgroup entered the kernel
    for tm in (0, 1):
        q = r[:, tm]
        print(f"   team {tm}: entry {np.median(q[:, 0]):6.1f} (max {q[:, 0].max():6.1f}) | loop start {np.median(q[:, 1]):6.1f} | "
              f"loop end {np.median(q[:, 2]):6.1f} (min {q[:, 2].min():6.1f} max {q[:, 2].max():6.1f}) | stores done {np.median(q[:, 3]):6.1f} (max {q[:, 3].max():6.1f}) us")
    print(f"{name:6s} cycles per tile: slab-wait {med[0]:7.0f} | halo->LDS {med[1]:7.0f} | prefetch issue {med[4]:7.0f} | MFMA {med[2]:7.0f} | epilogue {med[3]:7.0f} | sum {med.sum():7.0f}", flush=True)
