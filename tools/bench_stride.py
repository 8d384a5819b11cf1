#!/usr/bin/env python3
"""What the channel-slice layout of the concat buffers costs the streaming kernels: bn_act (no pool / with pool) at 64 ch x
256^2, B = 32, bf16, writing its output densely ([M, 64]) or into the skip half of a concat buffer ([M, 128], every other 128
bytes), and the transposed convolution 128 -> 64 @128^2 writing densely or into the up half."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr

P = hip.PREC_NAMES["bf16"]
B, H, C = 32, 256, 64
M = B * H * H


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


y = torch.randn(M, C, device="cuda").bfloat16()
v = torch.rand(2, C, device="cuda") + 0.5
dense = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
cat = torch.empty(M, 2 * C, device="cuda", dtype=torch.bfloat16)
pool = torch.empty(M // 4, C, device="cuda", dtype=torch.bfloat16)
for tag, out, ld in (("dense [M,64]", dense, C), ("skip half of [M,128]", cat[:, C:], 2 * C)):
    t = timeit(lambda: call("crimac_bn_act_pool", P, ptr(y), C, ptr(v[0]), ptr(v[1]), 1, out.data_ptr(), ld, None, 0, B, H, H, C))
    print(f"bn_act       -> {tag:22s} {t:7.1f} us  {2 * M * C * 2 / t / 1e6:5.2f} TB/s")
    t = timeit(lambda: call("crimac_bn_act_pool", P, ptr(y), C, ptr(v[0]), ptr(v[1]), 1, out.data_ptr(), ld, ptr(pool), C, B, H, H, C))
    print(f"bn_act_pool  -> {tag:22s} {t:7.1f} us  {2.25 * M * C * 2 / t / 1e6:5.2f} TB/s")
# reading a strided half (what unpool_add / the transposed convolution's input gradient do): bn_act FROM the half
for tag, src, ld in (("dense [M,64]", y, C), ("up half of [M,128]", cat[:, :C], 2 * C)):
    t = timeit(lambda: call("crimac_bn_act_pool", P, src.data_ptr(), ld, ptr(v[0]), ptr(v[1]), 1, ptr(dense), C, None, 0, B, H, H, C))
    print(f"bn_act     from {tag:22s} {t:7.1f} us  {2 * M * C * 2 / t / 1e6:5.2f} TB/s")
# transposed convolution 128 -> 64, 128^2 -> 256^2
h, Ci, Co = 128, 128, 64
Mi = B * h * h
x = torch.randn(Mi, Ci, device="cuda").bfloat16()
wf = torch.randint(-3000, 3000, (2 * 4 * Ci * Co,), dtype=torch.int16, device="cuda")
bias = torch.randn(Co, device="cuda")
for tag, out, ld in (("dense [4M,64]", dense, Co), ("up half of [4M,128]", cat, 2 * Co)):
    t = timeit(lambda: call("crimac_igemm_conv", P, ptr(x), Ci, B, h, h, h, h, Ci, 4 * Co, 1, 1, 0, 1, ptr(wf), ptr(wf), ptr(bias), Co,
                            out.data_ptr(), ld, 0, 1, Co))
    print(f"upconv 128->64 -> {tag:22s} {t:7.1f} us  {(Mi * Ci + 4 * Mi * Co) * 2 / t / 1e6:5.2f} TB/s")
