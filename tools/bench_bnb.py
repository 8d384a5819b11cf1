#!/usr/bin/env python3
"""bn_bwd_apply at the U-Net's shapes (B=32, bf16): streaming form vs the colreduce form (CRIMAC_BNB_STREAM=0)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr

P = hip.PREC_NAMES["bf16"]
tot = 0.0
for C, H, n in ((64, 256, 4), (128, 128, 4), (256, 64, 4), (512, 32, 4), (1024, 16, 2)):
    M = 32 * H * H
    da = torch.randn(M, C, device="cuda").bfloat16(); y = torch.randn(M, C, device="cuda").bfloat16()
    dy = torch.empty_like(da)
    v = torch.rand(4, C, device="cuda") + 0.5
    s = torch.randn(2, C, device="cuda", dtype=torch.float64)
    g = torch.zeros(2, C, device="cuda")
    fn = lambda: call("crimac_bn_bwd_apply", P, ptr(da), C, ptr(y), C, ptr(v[2]), ptr(v[3]), ptr(v[0]), ptr(v[1]),
                      ptr(s[0]), ptr(s[1]), M, 0, C, ptr(dy), C, ptr(g[0]), ptr(g[1]), None)
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    tot += n * us
    print(f"C={C:5d} @{H:3d}: {us:7.1f} us  {3 * M * C * 2 / us / 1e6:6.2f} TB/s")
print(f"per step (18 launches): {tot:.0f} us")
