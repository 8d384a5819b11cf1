#!/usr/bin/env python3
"""Micro-benchmark of the encoder-level backward edge at B = 32: crimac_unpool_add (storing da / sums only) and
crimac_unpool_bn_bwd_apply_replicas, per level, with the HBM floor of each.  usage: bench_unpool.py [bf16|h3f]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
fwd = hip.PREC_H3P if prec == "h3f" else hip.PREC_NAMES[prec]
app = hip.PREC_H3F_BWD if prec == "h3f" else hip.PREC_NAMES[prec]
dt = torch.float32 if prec == "h3f" else torch.bfloat16
dyt = torch.float16 if prec == "h3f" else torch.bfloat16
es, eo = (4, 2) if prec == "h3f" else (2, 2)


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


B = 32
for lvl, (H, C) in enumerate([(256, 64), (128, 128), (64, 256), (32, 512)]):
    M, Mp = B * H * H, B * (H // 2) ** 2
    y = torch.randn(M, C, device="cuda").to(dt)
    dp = torch.randn(Mp, C, device="cuda").to(dt)
    ds = torch.randn(M, C, device="cuda").to(dt)
    da = torch.empty(M, C, device="cuda", dtype=dt)
    dy = torch.empty(M, C, device="cuda", dtype=dyt)
    vec = torch.rand(4, 2048, device="cuda") + 0.5
    nrep = {64: 64, 128: 32, 256: 16, 512: 8}[C]          # engine._nrep
    s0 = torch.zeros(nrep, C, dtype=torch.float64, device="cuda")
    s1 = torch.zeros(nrep, C, dtype=torch.float64, device="cuda")
    dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    dummy = torch.zeros(8, device="cuda")
    t_store = timeit(lambda: call("crimac_unpool_add", fwd, ptr(dp), C, ptr(dummy), C, ptr(ds), C, ptr(da), C, B, H, H, C,
                                  ptr(y), C, ptr(vec), 2048, ptr(s0), ptr(s1), nrep))
    t_sums = timeit(lambda: call("crimac_unpool_add", fwd, ptr(dp), C, ptr(dummy), C, ptr(ds), C, None, 0, B, H, H, C,
                                 ptr(y), C, ptr(vec), 2048, ptr(s0), ptr(s1), nrep))
    t_apply = timeit(lambda: call("crimac_bn_bwd_apply_replicas", app, ptr(da), C, ptr(y), C, ptr(vec), 2048, ptr(s0), ptr(s1),
                                  nrep, M, M, C, ptr(dy), C, ptr(dg), ptr(db)))
    t_fused = timeit(lambda: call("crimac_unpool_bn_bwd_apply_replicas", app, ptr(dp), C, ptr(ds), C, ptr(y), C, ptr(vec), 2048,
                                  ptr(s0), ptr(s1), nrep, M, ptr(dy), C, B, H, H, C, ptr(dg), ptr(db)))
    n = M * C
    rd = 2.25 * es * n
    fl = lambda byts: byts / 5.2e12 * 1e6
    print(f"level {lvl} ({C} ch @ {H}^2): unpool_add storing da {t_store:7.1f} us (floor {fl(rd + es * n):6.1f}), sums only {t_sums:7.1f} "
          f"(floor {fl(rd):6.1f}); apply {t_apply:7.1f} (floor {fl(2 * es * n + eo * n):6.1f}), fused apply {t_fused:7.1f} "
          f"(floor {fl(rd + eo * n):6.1f})", flush=True)
