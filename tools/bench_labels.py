#!/usr/bin/env python3
"""Time of the on-GPU augmentation + label transform for one B=32 batch of 4x256x256 crops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crimac_classifiers_unet_amd as pkg

rng = np.random.default_rng(0)
B, H, W = 32, 256, 256
data = torch.from_numpy(np.power(10.0, rng.uniform(-9, -2, (B, 4, H, W))).astype(np.float32)).cuda()
lab = np.zeros((B, H, W), dtype=np.int16)
yy, xx = np.mgrid[0:H, 0:W]
for b in range(B):
    for k in range(6):
        cy, cx = rng.integers(0, H), rng.integers(0, W)
        lab[b][((yy - cy) / rng.integers(5, 40)) ** 2 + ((xx - cx) / rng.integers(5, 60)) ** 2 <= 1] = [27, 1, 12][k % 3]
lab = torch.from_numpy(lab).cuda()
m = pkg.UNet_Baseline(3, 4, precision="bf16").cuda()
for name, kw in (("augment + dB", {}), ("augment + dB + label transform", {"refine_labels": (3, 1e-7, 1e-4)})):
    for _ in range(3):
        m.engine.augment_batch(data, lab, 1, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(20):
        m.engine.augment_batch(data, lab, i, **kw)
    e.record()
    torch.cuda.synchronize()
    print(f"{name:34s} {1e3 * s.elapsed_time(e) / 20:8.1f} us per batch of {B}")
