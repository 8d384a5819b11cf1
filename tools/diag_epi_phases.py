#!/usr/bin/env python3
"""Where the shared convolution epilogue spends its cycles (diagnostic build of conv3x3_glds.hip with -DCRIMAC_DIAG_EPI,
CRIMAC_LIB selects it): wave 0 of each workgroup stamps s_memtime at the phase boundaries of conv_epilogue_body."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr
lib = hip.load_library()
rd = lib.crimac_diag_epi_read; rd.argtypes = [C.c_void_p]; rd.restype = C.c_int
PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, P = 32, hip.PREC_NAMES[PREC]
HPM = PREC == "h3p"
for name, H, Ci, Co in [("e1c1 64->128@128", 128, 64, 128), ("e1c2 128->128@128", 128, 128, 128), ("e2c2 256->256@64", 64, 256, 256)]:
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
    for mode in (0, 1):
        fn = lambda: call("crimac_conv3x3", P, ptr(x), Ci, B, H, H, Ci, Co, ptr(w), ptr(w), ptr(bias), ptr(out), Co, 0, mode,
                          ptr(st[0]) if mode else None, ptr(st[1]) if mode else None, 64, None, 0, None, 0)
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (1024 * 8))(); assert rd(buf) == 0
        a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(1024, 8)
        nwg = min(1024, B * (H // 16) ** 2 * max(Co // 128, 1))
        a = a[:nwg]
        d = np.diff(a[:, :6], axis=1)
        m = np.median(d, axis=0)
        print(f"{PREC} {name:20s} mode {mode}: entry->slice 0 staged {m[0]:6.0f} | barrier {m[1]:6.0f} | stores of slice 0 issued {m[2]:6.0f} | "
              f"remaining slices {m[3]:6.0f} | statistics flush {m[4]:6.0f} | total {m.sum():6.0f}", flush=True)
