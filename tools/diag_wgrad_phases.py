#!/usr/bin/env python3
"""Where a tile of wgrad_kernel spends its cycles (diagnostic build -DCRIMAC_DIAG_CLOCK -DCRIMAC_DIAG_PHASES,
CRIMAC_LIB selects it): s_memtime at the phase boundaries, waves 0 (taps 0-4) and 2 (taps 5-8) of each workgroup."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr
lib = hip.load_library()
rd = lib.crimac_diag_clock_wgrad_read; rd.argtypes = [C.c_void_p]; rd.restype = C.c_int
B, P = 32, hip.PREC_NAMES["bf16"]
for name, H, Ci, Co in [("e0c2 64->64@256", 256, 64, 64), ("e1c2 128->128@128", 128, 128, 128), ("e2c2 256->256@64", 64, 256, 256),
                        ("e3c2 512->512@32", 32, 512, 512), ("d0c1 1024->512@32", 32, 1024, 512)]:
    M = B * H * H
    x = torch.randn(M, Ci, device="cuda").bfloat16(); dy = torch.randn(M, Co, device="cuda").bfloat16()
    dw = torch.zeros(9 * Co * Ci, dtype=torch.float32, device="cuda")
    fn = lambda: call("crimac_wgrad", P, 0, ptr(dy), Co, Co, ptr(x), Ci, Ci, B, H, H, ptr(dw), 0)
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:20s} launch-to-launch {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us (diagnostic build)")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:20s} launch-to-launch {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us (diagnostic build)")
    buf = (C.c_ulonglong * (2 * 4096))(); assert rd(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 2).astype(np.float64)[:3072].reshape(256, 2, 2, 3, 2)   # wg, team, tap group, phase
    per = a[..., 0] / np.maximum(a[..., 1], 1)
    print("    tiles taken per team (median): team 0", np.median(a[:, 0, 0, 0, 1]), "team 1", np.median(a[:, 1, 0, 0, 1]))
    raw = np.frombuffer(buf, dtype=np.uint64)[6144:6144 + 1024].astype(np.int64).reshape(256, 4)
    r = (raw - raw[:, 0].min()) / 100.0
    print(f"{name:20s} us since the first workgroup entered: entry med {np.median(r[:, 0]):5.1f} max {r[:, 0].max():5.1f} | loop start "
          f"{np.median(r[:, 1]):5.1f} | loop end med {np.median(r[:, 2]):6.1f} min {r[:, 2].min():6.1f} max {r[:, 2].max():6.1f} | atomics issued med "
          f"{np.median(r[:, 3]):6.1f} max {r[:, 3].max():6.1f}")
    h, edges = np.histogram(r[:, 2], bins=12)
    print("    loop-end histogram (us):", " ".join(f"{e:.0f}:{c}" for e, c in zip(edges[:-1], h)))
    raw = np.frombuffer(buf, dtype=np.uint64)[6144:6144 + 1024].astype(np.int64).reshape(256, 4)
    r = (raw - raw[:, 0].min()) / 100.0
    print(f"{name:20s} us since the first workgroup entered: entry med {np.median(r[:, 0]):5.1f} max {r[:, 0].max():5.1f} | loop start "
          f"{np.median(r[:, 1]):5.1f} | loop end med {np.median(r[:, 2]):6.1f} min {r[:, 2].min():6.1f} max {r[:, 2].max():6.1f} | atomics issued med "
          f"{np.median(r[:, 3]):6.1f} max {r[:, 3].max():6.1f}")
    for team, tg in ((0, 0), (0, 1), (1, 0), (1, 1)):
        m = np.median(per[:, team, tg], axis=0)
        print(f"{name:20s} team {team} tap group {tg}:  cycles per tile: wait {m[0]:6.0f} | issue {m[1]:6.0f} | "
              f"contract {m[2]:6.0f} | sum {m.sum():6.0f}   (MFMA-bound: {(5 - tg) * 4 * 8 * 16 * 2} with two waves per SIMD)", flush=True)
