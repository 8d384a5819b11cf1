#!/usr/bin/env python3
"""64 -> 64 @256x256 convolution (persistent kernel) in its four epilogue forms; A/B between library builds with
CRIMAC_LIB.  usage: python tools/bench_p64.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PN = sys.argv[2] if len(sys.argv) > 2 else "bf16"
P = hip.PREC_NAMES[PN]
DT = torch.bfloat16 if PN == "bf16" else torch.float16
H = 256; C = 64; M = B * H * H
x = torch.randn(M, C, device="cuda").to(DT)
y = torch.randn(M, C, device="cuda").to(DT)
w = torch.randint(-3000, 3000, (9 * C * C,), dtype=torch.int16, device="cuda")
bias = torch.randn(C, device="cuda")
out = torch.empty(M, C, device="cuda", dtype=DT)
pool = torch.empty(M // 4, C, device="cuda", dtype=DT)
st = torch.zeros(2, 64, C, dtype=torch.float64, device="cuda")
vec = torch.rand(4, C, device="cuda") + 0.5
forms = {
    "plain": lambda: call("crimac_conv3x3", P, ptr(x), C, B, H, H, C, C, ptr(w), ptr(w), ptr(bias), ptr(out), C, 1, 0,
                          None, None, 64, None, 0, None, 0),
    "stats": lambda: call("crimac_conv3x3", P, ptr(x), C, B, H, H, C, C, ptr(w), ptr(w), ptr(bias), ptr(out), C, 0, 1,
                          ptr(st[0]), ptr(st[1]), 64, None, 0, None, 0),
    "bnb": lambda: call("crimac_conv3x3", P, ptr(x), C, B, H, H, C, C, ptr(w), ptr(w), None, ptr(out), C, 0, 2,
                        ptr(st[0]), ptr(st[1]), 64, ptr(y), C, ptr(vec), C),
}
if hasattr(hip.load_library(), "crimac_conv3x3_pool"):
    forms["pool"] = lambda: call("crimac_conv3x3_pool", P, ptr(x), C, B, H, H, C, C, ptr(w), ptr(w), ptr(bias), ptr(out),
                                 C, 1, ptr(pool), C)
flops = 2.0 * 9 * C * C * M
for name, fn in forms.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"B={B} {PN} {name:6s} {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {(2 + (name == 'bnb')) * M * C * 2 / us / 1e6:5.2f} TB/s", flush=True)
