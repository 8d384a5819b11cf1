#!/usr/bin/env python3
"""Micro-benchmark of the dense kernels on the U-Net's layer shapes (B=32, bf16 by default).

usage: python tools/bench_conv.py [conv|wgrad|all] [--prec bf16|f32x3] [--iters N] [--layers i,j,...]
Prints per-layer time and TFLOP/s (HIP events around `iters` back-to-back launches).
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr

# (name, H(=W), Cin, Cout) of the distinct 3x3 conv shapes of the baseline net
LAYERS = [("e0c2 64->64@256", 256, 64, 64), ("e1c1 64->128@128", 128, 64, 128),
          ("e1c2 128->128@128", 128, 128, 128), ("e2c1 128->256@64", 64, 128, 256),
          ("e2c2 256->256@64", 64, 256, 256), ("e3c1 256->512@32", 32, 256, 512),
          ("e3c2 512->512@32", 32, 512, 512), ("e4c1 512->1024@16", 16, 512, 1024),
          ("e4c2 1024->1024@16", 16, 1024, 1024), ("d0c1 1024->512@32", 32, 1024, 512),
          ("d1c1 512->256@64", 64, 512, 256), ("d2c1 256->128@128", 128, 256, 128),
          ("d3c1 128->64@256", 256, 128, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--layers", default="")
    ap.add_argument("--impl", default="halo")
    ap.add_argument("--nostat", action="store_true", help="conv without the fused BatchNorm statistics")
    a = ap.parse_args()
    P = hip.PREC_NAMES[a.prec]
    dt = torch.bfloat16 if a.prec == "bf16" else (torch.float16 if a.prec == "fp16" else torch.float32)
    hp = a.prec == "h3p"

    def act(M, C):
        """random activations in the precision's storage format (h3p: fp16 plane pairs [8 hi | 8 lo] per 8 channels)"""
        v = torch.randn(M, C, device="cuda")
        if not hp:
            return v.to(dt)
        hi = v.half()
        lo = (v - hi.float()).half()
        return torch.stack([hi.view(M, C // 8, 8), lo.view(M, C // 8, 8)], 2).contiguous().view(torch.float32).view(M, C)
    sel = [int(i) for i in a.layers.split(",")] if a.layers else range(len(LAYERS))
    B = a.batch
    tot_c = tot_w = 0.0
    for li in sel:
        name, H, Ci, Co = LAYERS[li]
        M = B * H * H
        x = act(M, Ci)
        dy = act(M, Co)
        w_hi = torch.randint(-3000, 3000, ((2 if hp else 1) * 9 * Co * Ci,), dtype=torch.int16, device="cuda")
        w_lo = torch.randint(-3000, 3000, (9 * Co * Ci,), dtype=torch.int16, device="cuda")
        bias = torch.randn(Co, device="cuda")
        out = torch.empty(M, Co, device="cuda", dtype=dt)
        stats = torch.zeros(2, 64, Co, dtype=torch.float64, device="cuda")
        dw = torch.zeros(9 * Co * Ci, dtype=torch.float32, device="cuda")
        flops = 2.0 * 9 * Ci * Co * M

        def conv():
            if a.impl == "halo":
                call("crimac_conv3x3", P, ptr(x), Ci, B, H, H, Ci, Co, ptr(w_hi), ptr(w_lo), ptr(bias), ptr(out),
                     Co, 0, 0 if a.nostat else 1, ptr(stats[0]), ptr(stats[1]), 64, None, 0, None, 0)
            else:
                call("crimac_igemm_conv", P, ptr(x), Ci, B, H, H, H, H, Ci, Co, 9, 3, 1, 1, ptr(w_hi), ptr(w_lo),
                     ptr(bias), Co, ptr(out), Co, 0, 0, 0)

        def wgrad():
            call("crimac_wgrad", P, 0, ptr(dy), Co, Co, ptr(x), Ci, Ci, B, H, H, ptr(dw), 0)

        for nm, fn in (("conv", conv), ("wgrad", wgrad)):
            if a.what not in ("all", nm):
                continue
            fn(); fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            us = 1e3 * s.elapsed_time(e) / a.iters
            if nm == "conv":
                tot_c += us
            else:
                tot_w += us
            print(f"{nm:5s} {name:22s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
    print(f"total conv {tot_c:.0f} us, wgrad {tot_w:.0f} us")
    if a.what in ("all", "upconv"):
        tot = 0.0
        for (h, Ci, Co) in ((16, 1024, 512), (32, 512, 256), (64, 256, 128), (128, 128, 64)):
            M = B * h * h
            x = torch.randn(M, Ci, device="cuda").to(dt)
            dy = torch.randn(4 * M, Co, device="cuda").to(dt)
            wf = torch.randint(-3000, 3000, (2 * 4 * Ci * Co,), dtype=torch.int16, device="cuda")
            bias = torch.randn(Co, device="cuda")
            out = torch.empty(4 * M, 2 * Co, device="cuda", dtype=dt)
            dx = torch.empty(M, Ci, device="cuda", dtype=dt)
            flops = 2.0 * 4 * Ci * Co * M

            def fwd():
                call("crimac_igemm_conv", P, ptr(x), Ci, B, h, h, h, h, Ci, 4 * Co, 1, 1, 0, 1, ptr(wf), ptr(wf),
                     ptr(bias), Co, ptr(out), 2 * Co, 0, 1, Co)

            def dgr():
                call("crimac_igemm_conv", P, ptr(dy), Co, B, 2 * h, 2 * h, h, h, Co, Ci, 4, 2, 0, 2, ptr(wf), ptr(wf),
                     None, 0, ptr(dx), Ci, 0, 0, 0)

            for nm, fn in (("upfwd", fwd), ("updgr", dgr)):
                fn(); fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.iters):
                    fn()
                e.record()
                torch.cuda.synchronize()
                us = 1e3 * s.elapsed_time(e) / a.iters
                tot += us
                print(f"{nm:5s} {Ci}->{Co}@{h}          {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
        print(f"total upconv fwd+dgrad {tot:.0f} us")


if __name__ == "__main__":
    main()
