#!/usr/bin/env python3
"""Micro-benchmark of the level-0 pointwise kernels (head backward, first-layer conv / wgrad) at B=32."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr


def timeit(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    B, H, C, NC = a.batch, 256, 64, 3
    M = B * H * H
    P = hip.PREC_NAMES["bf16"]
    dl = torch.randn(B, NC, H, H, device="cuda")
    x = torch.randn(M, C, device="cuda").bfloat16()
    y = torch.randn(M, C, device="cuda").bfloat16()
    dx = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(NC, C, device="cuda")
    dw = torch.zeros(NC, C, device="cuda")
    db = torch.zeros(NC, device="cuda")
    vec = torch.randn(4, C, device="cuda")
    stats = torch.zeros(2, 64, C, dtype=torch.float64, device="cuda")

    def head_bwd():
        call("crimac_head_bwd", P, ptr(dl), ptr(x), C, C, ptr(w), ptr(dx), C, ptr(dw), ptr(db), B, H, H, NC,
             ptr(y), C, ptr(vec), C, ptr(stats[0]), ptr(stats[1]), 64)
    t = timeit(head_bwd, a.iters)
    print(f"head_bwd (+BN-bwd sums) 64ch@256   {t:8.1f} us   {3 * M * C * 2 / t / 1e6:6.2f} TB/s")

    logits = torch.empty(B, NC, H, H, device="cuda")
    bias3 = torch.randn(NC, device="cuda")

    def head_fwd():
        call("crimac_head_fwd", P, ptr(x), C, C, ptr(w), ptr(bias3), ptr(logits), B, H, H, NC, 1, ptr(vec[2]), ptr(vec[3]))
    t = timeit(head_fwd, a.iters)
    print(f"head_fwd (BN+ReLU+1x1+softmax) 64ch@256 {t:8.1f} us   {(M * C * 2 + M * NC * 4) / t / 1e6:6.2f} TB/s")

    # first layer: Cin 4 (padded to 16) -> 64
    Ci = 16
    x0 = torch.randn(M, Ci, device="cuda").bfloat16()
    out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    w_hi = torch.randint(-3000, 3000, (9 * C * Ci,), dtype=torch.int16, device="cuda")
    bias = torch.randn(C, device="cuda")

    def conv0():
        call("crimac_conv3x3", P, ptr(x0), Ci, B, H, H, Ci, C, ptr(w_hi), ptr(w_hi), ptr(bias), ptr(out), C, 0, 1,
             ptr(stats[0]), ptr(stats[1]), 64, None, 0, None, 0)
    t = timeit(conv0, a.iters)
    print(f"conv e0c1 16->64@256 (+stats)      {t:8.1f} us   {(M * C * 2 + M * Ci * 2) / t / 1e6:6.2f} TB/s")

    def conv0n():
        call("crimac_conv3x3", P, ptr(x0), Ci, B, H, H, Ci, C, ptr(w_hi), ptr(w_hi), ptr(bias), ptr(out), C, 0, 0,
             None, None, 1, None, 0, None, 0)
    t = timeit(conv0n, a.iters)
    print(f"conv e0c1 16->64@256 (no stats)    {t:8.1f} us   {(M * C * 2 + M * Ci * 2) / t / 1e6:6.2f} TB/s")

    dwg = torch.zeros(9 * C * Ci, device="cuda")

    def wgrad0():
        call("crimac_wgrad", P, 0, ptr(y), C, C, ptr(x0), Ci, Ci, B, H, H, ptr(dwg), 0)
    t = timeit(wgrad0, a.iters)
    print(f"wgrad e0c1 16->64@256              {t:8.1f} us   {(M * C * 2 + M * Ci * 2) / t / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
