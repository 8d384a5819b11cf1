#!/usr/bin/env python3
"""stat_mode 2 of the persistent 64-channel kernel vs crimac_bn_bwd_reduce: which channels differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_kernels import _round, to_nhwc, pack_conv, _dt
from crimac_classifiers_unet_amd import hip
from crimac_classifiers_unet_amd.hip import call, ptr
for prec in ("bf16", "fp16"):
    B, H, W, Ci, Co = 3, 256, 256, 64, 64
    g = torch.Generator().manual_seed(21)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    dy = _round(torch.randn(B, Co, H, W, generator=g), prec)
    y_prev = _round(torch.randn(B, Ci, H, W, generator=g) * 1.5 + 0.3, prec)
    mean, invstd = torch.randn(Ci, generator=g) * 0.2, torch.rand(Ci, generator=g) + 0.5
    scale, shift = (torch.rand(Ci, generator=g) + 0.5) * invstd, torch.randn(Ci, generator=g) * 0.3
    _, _, dh, dl = pack_conv(w, prec)
    M = B * H * W; R = 7
    dyn, yn = to_nhwc(dy, prec), to_nhwc(y_prev, prec)
    vec = torch.stack([mean, invstd, scale, shift]).contiguous().cuda()
    for rep in range(3):
        acc = torch.zeros(2, R, Ci, dtype=torch.float64, device="cuda")
        da = torch.empty(M, Ci, dtype=_dt(prec), device="cuda")
        call("crimac_conv3x3", hip.PREC_NAMES[prec], ptr(dyn), Co, B, H, W, Co, Ci, ptr(dh), ptr(dl), None, ptr(da), Ci,
             0, 2, ptr(acc[0]), ptr(acc[1]), R, ptr(yn), Ci, ptr(vec), Ci)
        ref = torch.zeros(2, Ci, dtype=torch.float64, device="cuda")
        call("crimac_bn_bwd_reduce", hip.PREC_NAMES[prec], ptr(da), Ci, ptr(yn), Ci, ptr(vec[2]), ptr(vec[3]),
             ptr(vec[0]), ptr(vec[1]), M, Ci, ptr(ref[0]), ptr(ref[1]))
        out = acc.sum(1)
        torch.cuda.synchronize()
        for k in (0, 1):
            rel = ((out[k] - ref[k]).abs() / ref[k].abs().max()).cpu()
            bad = (rel > 1e-5).nonzero().flatten().tolist()
            print(prec, "rep", rep, "sum", k, "bad channels", bad, "max rel", float(rel.max()))
