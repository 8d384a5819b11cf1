"""GPU parity tests of precision 'h3p' (CRIMAC_PREC_H3P: fp16 plane pairs split once by the producer, 3 MFMAs per
product), one kernel at a time through the C ABI, against the torch CPU fp32 op the reference dispatches for the same
step -- then the whole network against the reference golden.

A plane-pair tensor has fp32 addressing; every aligned group of 8 channels holds [8 x fp16 hi][8 x fp16 lo],
value = hi + lo (22 significant bits).  Inputs are pre-rounded to that format on the host, so what a kernel adds is
the dropped lo*lo products (~2^-22) and fp32 accumulation order: tolerance 5e-6 of max|ref| for one contraction.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import hip, synth
from crimac_classifiers_unet_amd.hip import call, ptr

pytestmark = pytest.mark.gpu

P = hip.PREC_H3P
TOL = 5e-6
WS = 2.0 ** 8                       # weight planes are pre-scaled by 2^CRIMAC_F32H3_WSHIFT, undone in the epilogues


def hp_round(x):
    """fp32 -> the value a plane pair holds for it (hi + lo, both fp16)."""
    hi = x.half()
    lo = (x - hi.float()).half()
    return hi.float() + lo.float()


def hp_pack(m):
    """[M, C] fp32 (cpu) -> [M, C] 'float32' tensor whose bytes are the plane pairs of m."""
    M, C = m.shape
    assert C % 8 == 0
    hi = m.half()
    lo = (m - hi.float()).half()
    both = torch.stack([hi.view(M, C // 8, 8), lo.view(M, C // 8, 8)], dim=2)      # [M, C/8, 2, 8] halves
    return both.contiguous().view(torch.float32).view(M, C).clone()


def hp_unpack(t):
    """inverse of hp_pack: [M, C] plane-pair bytes (any device) -> fp32 values (cpu)."""
    M, C = t.shape
    h = t.contiguous().cpu().view(torch.float16).view(M, C // 8, 2, 8).float()
    return (h[:, :, 0] + h[:, :, 1]).reshape(M, C)


def to_nhwc_hp(x, ld=None, off=0):
    """[B,C,H,W] fp32 (cpu) -> device [B*H*W, ld] buffer with the plane pairs of x in channels [off, off + C)."""
    B, C, H, W = x.shape
    ld = ld or C
    full = torch.zeros(B * H * W, ld, dtype=torch.float32)
    full[:, off:off + C] = x.permute(0, 2, 3, 1).reshape(-1, C)
    return hp_pack(full).cuda()                   # (zeros pack to zeros: the padding channels stay 0)


def from_nhwc(t, B, H, W):
    return t.float().cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


def from_nhwc_hp(t, B, H, W):
    return hp_unpack(t).reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def w_round(w):
    """What the packed fp16 weight planes represent: hp_round(w * 2^8) / 2^8."""
    return hp_round(w * WS) / WS


def pack_conv(w, cin_pad=None, scale=None, dgrad=True):
    Co, Ci = w.shape[:2]
    cin_pad = cin_pad or Ci
    wd = w.float().cuda().contiguous()
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.empty(2 * 9 * Co * cin_pad, **i16), torch.empty(8, **i16)
    dh = torch.empty(2 * 9 * Ci * Co, **i16) if dgrad and cin_pad == Ci else None
    dl = torch.empty(8, **i16) if dh is not None else None
    sc = scale.float().cuda() if scale is not None else None
    call("crimac_pack_conv3x3", ptr(wd), Co, Ci, cin_pad, ptr(sc), hip.PLANES_H3P, ptr(fh), ptr(fl), ptr(dh), ptr(dl))
    torch.cuda.synchronize()
    return fh, fl, dh, dl


def conv(x_hp, ld, B, H, W, Cin, Cout, w, wl, bias, relu=0, out_planes=False, stats=None, bnb=None, cols=None, out=None,
         out_ld=None, out_off=0, pool=None):
    out_ld = out_ld or Cout
    if out is None:
        out = torch.full((B * H * W, out_ld), 7.0, dtype=torch.float32, device="cuda")
    flags = (hip.EPI_RELU if relu else 0) | (hip.EPI_OUT_PLANES if out_planes else 0)
    mode, s0, s1, by, by_ld, bvec, bstride = 0, None, None, None, 0, None, 0
    if stats is not None:
        mode, s0, s1 = 1, ptr(stats[0]), ptr(stats[1])
    if bnb is not None:
        mode = 2
        y, vec, s0t, s1t = bnb
        s0, s1, by, by_ld, bvec, bstride = ptr(s0t), ptr(s1t), ptr(y), y.shape[1], ptr(vec), vec.shape[1]
    if pool is not None:
        call("crimac_conv3x3_pool", P, ptr(x_hp), ld, B, H, W, Cin, Cout, ptr(w), ptr(wl), ptr(bias), ptr(out, out_off), out_ld,
             flags, ptr(pool), pool.shape[1])
    elif cols is not None:
        call("crimac_conv3x3_cols", P, ptr(x_hp), ld, B, H, W, Cin, Cout, ptr(w), ptr(wl), ptr(bias), ptr(out, out_off), out_ld,
             flags, mode, s0, s1, 4, by, by_ld, bvec, bstride, cols[0], cols[1])
    else:
        call("crimac_conv3x3", P, ptr(x_hp), ld, B, H, W, Cin, Cout, ptr(w), ptr(wl), ptr(bias), ptr(out, out_off), out_ld,
             flags, mode, s0, s1, 4, by, by_ld, bvec, bstride)
    torch.cuda.synchronize()
    return out


def test_layout_conversion_writes_plane_pairs():
    B, C, H, W = 2, 4, 8, 16
    x = torch.randn(B, C, H, W) * 30
    out = torch.empty(B * H * W, 16, dtype=torch.float32, device="cuda")
    xd = x.cuda()
    call("crimac_nchw_to_nhwc", P, ptr(xd), ptr(out), B, C, H, W, 16)
    torch.cuda.synchronize()
    ref = torch.zeros(B * H * W, 16)
    ref[:, :C] = x.permute(0, 2, 3, 1).reshape(-1, C)
    assert torch.equal(out.cpu().view(torch.int32), hp_pack(ref).view(torch.int32))          # bit for bit
    assert relerr(hp_unpack(out), ref) < 2.0 ** -21


# (B, H, W, Cin, Cout): channel-split kernel (Cout % 128 == 0), pixel-split kernel (Cout = 64), first layer (Cin = 4 -> 16),
# partial tiles, many workgroups
CONV_SHAPES = [(2, 16, 16, 64, 128), (1, 16, 32, 128, 128), (3, 8, 8, 256, 64), (2, 16, 16, 4, 64), (1, 24, 40, 64, 192),
               (2, 21, 37, 32, 64), (1, 19, 23, 96, 256), (4, 64, 64, 128, 256), (8, 64, 64, 64, 64),
               (5, 250, 200, 32, 128), (3, 120, 136, 64, 256)]       # (channel-split kernel, more than two resident rounds, ragged)


@pytest.mark.parametrize("shape", [(2, 32, 48, 128, 128), (1, 24, 40, 32, 256), (3, 16, 16, 256, 128)])
def test_fragment_major_plane_pair_weights_convolve_bit_for_bit(shape):
    """CRIMAC_EPI_WFRAG on plane pairs (round 5): the interleaved rows of 2 Cin halves packed fragment-major -- by the per-layer
    entry point (CRIMAC_PLANES_FWD_FRAG, the eval packs) and by crimac_pack_layers (CRIMAC_LAYER_*_FRAG, the training packs) --
    give BIT-identical convolutions: forward with statistics to fp32, forward to plane pairs with the fused pool, input
    gradient with the fused BatchNorm-backward sums, and the h3f backward's fp16 input gradient."""
    import ctypes
    B, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(41)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    wd = w.cuda().contiguous()
    i16 = dict(dtype=torch.int16, device="cuda")
    n = 2 * 9 * Co * Ci
    lo = torch.zeros(8, **i16)
    planes = {}
    for tag, kind in (("row", 0), ("frag", hip.LAYER_FWD_FRAG | (hip.LAYER_DG_FRAG if Ci % 128 == 0 else 0))):
        d = (hip.LayerDesc * 1)()
        planes[tag] = {"fwd": torch.full((n,), -1, **i16), "dg": torch.full((n,), -1, **i16)}
        d[0].w = wd.data_ptr()
        d[0].fwd_hi, d[0].fwd_lo = planes[tag]["fwd"].data_ptr(), lo.data_ptr()
        d[0].dg_hi, d[0].dg_lo = planes[tag]["dg"].data_ptr(), lo.data_ptr()
        d[0].kind, d[0].Co, d[0].Ci, d[0].Ci_pad = kind, Co, Ci, Ci
        call("crimac_pack_layers", ctypes.byref(d), 1, hip.PLANES_H3P)
    one = torch.full((n,), -1, **i16)           # the per-layer entry point writes the same fragment-major forward plane
    call("crimac_pack_conv3x3", ptr(wd), Co, Ci, Ci, None, hip.PLANES_H3P | hip.PLANES_FWD_FRAG, ptr(one), ptr(lo), None, None)
    torch.cuda.synchronize()
    assert torch.equal(one, planes["frag"]["fwd"]) and not torch.equal(one, planes["row"]["fwd"])
    assert torch.equal(torch.sort(one).values, torch.sort(planes["row"]["fwd"]).values)          # a permutation of the halves
    x = to_nhwc_hp(torch.randn(B, Ci, H, W, generator=g))
    dy = to_nhwc_hp(torch.randn(B, Co, H, W, generator=g))
    bias = torch.randn(Co, generator=g).cuda()
    yprev = (torch.randn(B * H * W, Ci, generator=g) * 1.5 + 0.3).cuda()
    vec = torch.stack([torch.randn(Ci, generator=g) * 0.2, torch.rand(Ci, generator=g) + 0.5,
                       torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3]).contiguous().cuda()
    res = {}
    for tag, flag in (("row", 0), ("frag", hip.EPI_WFRAG)):
        st = torch.zeros(2, 4, Co, dtype=torch.float64, device="cuda")
        o32 = torch.full((B * H * W, Co), 7.0, dtype=torch.float32, device="cuda")
        call("crimac_conv3x3", P, ptr(x), Ci, B, H, W, Ci, Co, ptr(planes[tag]["fwd"]), ptr(lo), ptr(bias), ptr(o32), Co,
             hip.EPI_RELU | flag, 1, ptr(st[0]), ptr(st[1]), 4, None, 0, None, 0)
        opp = torch.full((B * H * W, Co), 7.0, dtype=torch.float32, device="cuda")
        pool = torch.full((B * (H // 2) * (W // 2), Co), 7.0, dtype=torch.float32, device="cuda")
        call("crimac_conv3x3_pool", P, ptr(x), Ci, B, H, W, Ci, Co, ptr(planes[tag]["fwd"]), ptr(lo), ptr(bias), ptr(opp), Co,
             hip.EPI_RELU | hip.EPI_OUT_PLANES | flag, ptr(pool), Co)
        da = None
        if Ci % 128 == 0:
            da = torch.full((B * H * W, Ci), 7.0, dtype=torch.float32, device="cuda")
            acc = torch.zeros(2, 4, Ci, dtype=torch.float64, device="cuda")
            call("crimac_conv3x3", P, ptr(dy), Co, B, H, W, Co, Ci, ptr(planes[tag]["dg"]), ptr(lo), None, ptr(da), Ci, flag, 2,
                 ptr(acc[0]), ptr(acc[1]), 4, ptr(yprev), Ci, ptr(vec), Ci)
        torch.cuda.synchronize()
        res[tag] = (o32, opp, pool, da)
    for a, b in zip(res["row"], res["frag"]):
        assert (a is None and b is None) or torch.equal(a, b)
    assert float(res["frag"][0].abs().max()) > 0
    # the h3f backward's fp16 personality (CRIMAC_PREC_H3F_BWD: fp16 dy and planes, fp32 da)
    if Ci % 128 == 0:
        dy16 = torch.randn(B * H * W, Co, generator=g).half().cuda()
        out = {}
        for tag, kind, flag in (("row", 0, 0), ("frag", hip.LAYER_DG_FRAG, hip.EPI_WFRAG)):
            d = (hip.LayerDesc * 1)()
            f16, g16 = torch.full((n // 2,), -1, **i16), torch.full((n // 2,), -1, **i16)
            d[0].w = wd.data_ptr()
            d[0].fwd_hi, d[0].fwd_lo, d[0].dg_hi, d[0].dg_lo = f16.data_ptr(), lo.data_ptr(), g16.data_ptr(), lo.data_ptr()
            d[0].kind, d[0].Co, d[0].Ci, d[0].Ci_pad = kind, Co, Ci, Ci
            call("crimac_pack_layers", ctypes.byref(d), 1, hip.PLANES_FP16)
            da = torch.full((B * H * W, Ci), 7.0, dtype=torch.float32, device="cuda")
            call("crimac_conv3x3", hip.PREC_H3F_BWD, ptr(dy16), Co, B, H, W, Co, Ci, ptr(g16), None, None, ptr(da), Ci, flag, 0,
                 None, None, 1, None, 0, None, 0)
            torch.cuda.synchronize()
            out[tag] = da
        assert torch.equal(out["row"], out["frag"]) and float(out["frag"].abs().max()) > 0


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_forward_fp32_and_plane_pair_outputs(shape):
    B, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(1)
    x = hp_round(torch.randn(B, Ci, H, W, generator=g) * 3)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    b = torch.randn(Co, generator=g)
    cin_pad = 16 if Ci < 16 else Ci
    fh, fl, _, _ = pack_conv(w, cin_pad, dgrad=False)
    ref = F.conv2d(x.double(), w_round(w).double(), b.double(), padding=1).float()
    xin = to_nhwc_hp(x, ld=cin_pad)
    bd = b.cuda()
    out = conv(xin, cin_pad, B, H, W, cin_pad, Co, fh, fl, bd)
    assert relerr(from_nhwc(out, B, H, W), ref) < TOL
    # ReLU + plane-pair output (what the next convolution of an inference pass reads)
    out = conv(xin, cin_pad, B, H, W, cin_pad, Co, fh, fl, bd, relu=1, out_planes=True)
    got = from_nhwc_hp(out, B, H, W)
    assert relerr(got, torch.relu(ref)) < TOL
    # fused BatchNorm statistics of the fp32 output
    st = [torch.zeros(4 * Co, dtype=torch.float64, device="cuda") for _ in range(2)]
    out = conv(xin, cin_pad, B, H, W, cin_pad, Co, fh, fl, bd, stats=st)
    y = from_nhwc(out, B, H, W).double()
    s1, s2 = st[0].view(4, Co).sum(0).cpu(), st[1].view(4, Co).sum(0).cpu()
    assert relerr(s1, y.sum((0, 2, 3))) < 1e-5 and relerr(s2, (y * y).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("shape", [(2, 16, 16, 4), (3, 21, 37, 4), (1, 40, 24, 3), (20, 128, 128, 4), (2, 256, 256, 4)])
def test_first_layer_on_the_persistent_pseudo_channel_kernel(shape):
    """CRIMAC_EPI_CIN4: the network input (<= 4 real channels of the padded 16) through the persistent first-layer kernel
    (hi / lo planes as sixteen pseudo-channels) == F.conv2d, == the generic kernel to rounding; fp32 output with fused
    statistics (training) and ReLU + plane-pair output (inference); partial tiles and persistent tile loops (> 768 tiles)."""
    B, H, W, Ci = shape
    Co = 64
    g = torch.Generator().manual_seed(2)
    x = hp_round(torch.randn(B, Ci, H, W, generator=g) * 3)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    b = torch.randn(Co, generator=g)
    fh, fl, _, _ = pack_conv(w, 16, dgrad=False)
    ref = F.conv2d(x.double(), w_round(w).double(), b.double(), padding=1).float()
    xin = to_nhwc_hp(x, ld=16)
    bd = b.cuda()
    M = B * H * W

    def run(flags, stats=None):
        out = torch.full((M, Co), 7.0, dtype=torch.float32, device="cuda")
        mode, s0, s1 = (1, ptr(stats[0]), ptr(stats[1])) if stats is not None else (0, None, None)
        call("crimac_conv3x3", P, ptr(xin), 16, B, H, W, 16, Co, ptr(fh), ptr(fl), ptr(bd), ptr(out), Co, flags, mode, s0, s1, 4,
             None, 0, None, 0)
        torch.cuda.synchronize()
        return out

    st = [torch.zeros(4 * Co, dtype=torch.float64, device="cuda") for _ in range(2)]
    out = run(hip.EPI_CIN4, st)
    y = from_nhwc(out, B, H, W)
    assert relerr(y, ref) < TOL
    assert relerr(from_nhwc(run(0), B, H, W), y) < 2e-6                      # the generic kernel
    yd = y.double()
    s1, s2 = st[0].view(4, Co).sum(0).cpu(), st[1].view(4, Co).sum(0).cpu()
    assert relerr(s1, yd.sum((0, 2, 3))) < 1e-5 and relerr(s2, (yd * yd).sum((0, 2, 3))) < 1e-5
    got = from_nhwc_hp(run(hip.EPI_CIN4 | hip.EPI_RELU | hip.EPI_OUT_PLANES), B, H, W)
    assert relerr(got, torch.relu(ref)) < TOL


def test_conv3x3_strided_io_and_channel_ranges():
    """Input and output in channel slices of wider buffers (the concat buffers), and crimac_conv3x3_cols: the two halves
    of a decoder input gradient, the first as plane pairs with column sums, the second as fp32 -- in ONE buffer."""
    B, H, W, Ci, Co = 2, 32, 32, 64, 256
    g = torch.Generator().manual_seed(2)
    x = hp_round(torch.randn(B, Ci, H, W, generator=g))
    w = torch.randn(Co, Ci, 3, 3, generator=g) / 24
    fh, fl, _, _ = pack_conv(w, dgrad=False)
    ref = F.conv2d(x.double(), w_round(w).double(), None, padding=1).float()
    big_in = to_nhwc_hp(x, ld=2 * Ci, off=Ci)
    out = torch.full((B * H * W, Co + 64), 7.0, dtype=torch.float32, device="cuda")
    st = [torch.zeros(4 * Co, dtype=torch.float64, device="cuda") for _ in range(2)]
    xs = big_in[:, Ci:]
    call("crimac_conv3x3_cols", P, ptr(big_in, Ci), 2 * Ci, B, H, W, Ci, Co, ptr(fh), ptr(fl), None, ptr(out, 32), Co + 64,
         hip.EPI_OUT_PLANES, 1, ptr(st[0]), ptr(st[1]), 4, None, 0, None, 0, 0, 128)
    call("crimac_conv3x3_cols", P, ptr(big_in, Ci), 2 * Ci, B, H, W, Ci, Co, ptr(fh), ptr(fl), None, ptr(out, 32), Co + 64,
         0, 0, None, None, 4, None, 0, None, 0, 128, 128)
    torch.cuda.synchronize()
    del xs
    first = from_nhwc_hp(out[:, 32:32 + 128].contiguous(), B, H, W)
    second = from_nhwc(out[:, 32 + 128:32 + 256].contiguous(), B, H, W)
    assert relerr(first, ref[:, :128]) < TOL and relerr(second, ref[:, 128:]) < TOL
    assert float((out[:, :32] - 7).abs().max()) == 0 and float((out[:, 32 + 256:] - 7).abs().max()) == 0
    s1 = st[0].view(4, Co).sum(0).cpu()
    assert relerr(s1[:128], first.double().sum((0, 2, 3))) < 1e-5 and float(s1[128:].abs().max()) == 0


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 128), (1, 32, 32, 128, 64), (2, 32, 32, 64, 64)])
def test_conv3x3_dgrad_with_fused_bn_backward_sums(shape):
    """Input gradient (dgrad planes, fp32 output `da`) with the BatchNorm-backward sums of the block it feeds."""
    B, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(3)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    dy = hp_round(torch.randn(B, Co, H, W, generator=g) * 50)
    _, _, dh, dl = pack_conv(w)
    ref = torch.nn.grad.conv2d_input((B, Ci, H, W), w_round(w).double(), dy.double(), padding=1).float()
    dyn = to_nhwc_hp(dy)
    y = torch.randn(B * H * W, Ci, generator=g)
    vec = torch.stack([torch.randn(Ci, generator=g) * 0.1, torch.rand(Ci, generator=g) + 0.5,
                       torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3])      # mean, invstd, scale, shift
    yd, vd = y.cuda(), vec.cuda().contiguous()
    st = [torch.zeros(4 * Ci, dtype=torch.float64, device="cuda") for _ in range(2)]
    out = conv(dyn, Co, B, H, W, Co, Ci, dh, dl, None, bnb=(yd, vd, st[0], st[1]))
    da = from_nhwc(out, B, H, W)
    assert relerr(da, ref) < TOL
    dam = out.cpu().double()
    mask = (y.double() * vec[2].double() + vec[3].double()) > 0
    dz = torch.where(mask, dam, torch.zeros_like(dam))
    xh = (y.double() - vec[0].double()) * vec[1].double()
    assert relerr(st[0].view(4, Ci).sum(0).cpu(), dz.sum(0)) < 1e-5
    assert relerr(st[1].view(4, Ci).sum(0).cpu(), (dz * xh).sum(0)) < 1e-5


def test_conv3x3_with_fused_maxpool_plane_pairs():
    B, H, W, C = 2, 32, 32, 128
    g = torch.Generator().manual_seed(4)
    x = hp_round(torch.randn(B, C, H, W, generator=g))
    w = torch.randn(C, C, 3, 3, generator=g) / 34
    b = torch.randn(C, generator=g) * 0.1
    fh, fl, _, _ = pack_conv(w, dgrad=False)
    ref = torch.relu(F.conv2d(x.double(), w_round(w).double(), b.double(), padding=1)).float()
    xin, bd = to_nhwc_hp(x), b.cuda()
    pool = torch.empty(B * H * W // 4, C, dtype=torch.float32, device="cuda")
    out = conv(xin, C, B, H, W, C, C, fh, fl, bd, relu=1, out_planes=True, pool=pool)
    a = from_nhwc_hp(out, B, H, W)
    assert relerr(a, ref) < TOL
    assert torch.equal(from_nhwc_hp(pool, B, H // 2, W // 2), F.max_pool2d(a, 2))     # pool of the values as stored, exactly


@pytest.mark.parametrize("shape", [(2, 8, 8, 128, 64), (1, 4, 8, 256, 128), (2, 24, 40, 128, 64), (2, 64, 64, 256, 128),
                                   (3, 5, 7, 128, 64), (32, 16, 16, 1024, 512), (1, 8, 8, 384, 128)])
def test_upconv2x2_forward_dgrad_wgrad(shape):
    B, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(5)
    x = hp_round(torch.randn(B, Ci, H, W, generator=g))
    w = torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5
    b = torch.randn(Co, generator=g)
    i16 = dict(dtype=torch.int16, device="cuda")
    n = 4 * Ci * Co
    fh, dh = torch.empty(2 * n, **i16), torch.empty(2 * n, **i16)
    fl, dl = torch.empty(8, **i16), torch.empty(8, **i16)
    wd, bd, xn = w.cuda(), b.cuda(), to_nhwc_hp(x)
    call("crimac_pack_upconv2x2", ptr(wd), Ci, Co, hip.PLANES_H3P, ptr(fh), ptr(fl), ptr(dh), ptr(dl))
    wr = w_round(w)
    ref = F.conv_transpose2d(x.double(), wr.double(), b.double(), stride=2).float()
    cat = torch.zeros(B * 4 * H * W, 2 * Co, dtype=torch.float32, device="cuda")
    call("crimac_igemm_conv", P, ptr(xn), Ci, B, H, W, H, W, Ci, 4 * Co, 1, 1, 0, 1, ptr(fh), ptr(fl), ptr(bd), Co,
         ptr(cat), 2 * Co, hip.EPI_OUT_PLANES, 1, Co)
    torch.cuda.synchronize()
    assert relerr(from_nhwc_hp(cat[:, :Co].contiguous(), B, 2 * H, 2 * W), ref) < TOL
    assert float(cat[:, Co:].abs().max()) == 0
    # input gradient (fp32) with the fused BatchNorm-backward sums, and the plain form
    dy = hp_round(torch.randn(B, Co, 2 * H, 2 * W, generator=g) * 20)
    xg = x.double().clone().requires_grad_(True)
    wg = wr.double().clone().requires_grad_(True)
    F.conv_transpose2d(xg, wg, None, stride=2).backward(dy.double())
    dyn = to_nhwc_hp(dy)
    dx = torch.empty(B * H * W, Ci, dtype=torch.float32, device="cuda")
    call("crimac_igemm_conv", P, ptr(dyn), Co, B, 2 * H, 2 * W, H, W, Co, Ci, 4, 2, 0, 2, ptr(dh), ptr(dl), None, 0,
         ptr(dx), Ci, 0, 0, 0)
    torch.cuda.synchronize()
    assert relerr(from_nhwc(dx, B, H, W), xg.grad.float()) < TOL
    y = torch.randn(B * H * W, Ci, generator=g)
    vec = torch.stack([torch.randn(Ci, generator=g) * 0.1, torch.rand(Ci, generator=g) + 0.5,
                       torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3])
    yd, vd = y.cuda(), vec.cuda().contiguous()
    st = [torch.zeros(4 * Ci, dtype=torch.float64, device="cuda") for _ in range(2)]
    dx2 = torch.empty_like(dx)
    call("crimac_upconv2x2_dgrad_bnb_prec", P, ptr(dyn), Co, B, H, W, Co, Ci, ptr(dh), ptr(dx2), Ci, ptr(yd), Ci, ptr(vd), Ci,
         ptr(st[0]), ptr(st[1]), 4)
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx)
    dam = dx.cpu().double()
    dz = torch.where((y.double() * vec[2].double() + vec[3].double()) > 0, dam, torch.zeros_like(dam))
    assert relerr(st[0].view(4, Ci).sum(0).cpu(), dz.sum(0)) < 1e-5
    # weight gradient: F = x (plane pairs), S = dY (plane pairs)
    dwp = torch.zeros(n, dtype=torch.float32, device="cuda")
    call("crimac_wgrad", P, 1, ptr(xn), Ci, Ci, ptr(dyn), Co, Co, B, H, W, ptr(dwp), 2)
    grad = torch.empty(Ci, Co, 2, 2, dtype=torch.float32, device="cuda")
    call("crimac_unpack_wgrad_upconv2x2", ptr(dwp), Ci, Co, ptr(grad))
    torch.cuda.synchronize()
    assert relerr(grad.cpu(), wg.grad.float()) < 2 * TOL


# (B, H, W, Ci, Co): the 8-wave plane-pair kernel (Ci, Co multiples of 64), several channel tiles, partial tiles at the image
# border, many tiles per workgroup; and the shapes it does not take (first layer: Ci = 16; Co = 32)
@pytest.mark.parametrize("target", [0, 2, 24])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 32, 32, 128, 64), (2, 21, 37, 64, 128), (2, 64, 64, 128, 128),
                                   (4, 128, 128, 64, 64), (2, 16, 16, 16, 64), (2, 16, 16, 64, 32)])
def test_wgrad_conv3x3(shape, target):
    B, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(6)
    x = hp_round(torch.randn(B, Ci, H, W, generator=g))
    dy = hp_round(torch.randn(B, Co, H, W, generator=g) * 40)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 3, 3), dy.double(), padding=1).float()
    dwp = torch.zeros(9 * Co * Ci, dtype=torch.float32, device="cuda")
    dyn, xn = to_nhwc_hp(dy), to_nhwc_hp(x)
    call("crimac_wgrad", P, 0, ptr(dyn), Co, Co, ptr(xn), Ci, Ci, B, H, W, ptr(dwp), target)
    grad = torch.empty(Co, Ci, 3, 3, dtype=torch.float32, device="cuda")
    call("crimac_unpack_wgrad_conv3x3", ptr(dwp), Co, Ci, Ci, ptr(grad))
    torch.cuda.synchronize()
    assert relerr(grad.cpu(), ref) < 2 * TOL


@pytest.mark.parametrize("items", [0, 16])
def test_grouped_weight_gradients_plane_pairs(items):
    """crimac_wgrad_group in the plane-pair precision: several conv3x3 layers (full tiles, ragged images, an operand in a
    channel slice of a wider buffer, few-split layers) through one persistent launch == torch's conv2d_weight in fp64."""
    import ctypes
    lib = hip.load_library()
    g = torch.Generator().manual_seed(43)
    B = 2
    specs = [(128, 128, 64, 64, 0), (100, 120, 64, 128, 0), (20, 24, 128, 64, 256), (16, 16, 256, 128, 0), (32, 32, 512, 256, 0)]
    keep, arr = [], (hip.WgradGroupLayer * len(specs))()
    for d, (H, W, Ci, Co, ldx) in zip(arr, specs):
        x = hp_round(torch.randn(B, Ci, H, W, generator=g))
        dy = hp_round(torch.randn(B, Co, H, W, generator=g) * 40)
        ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 3, 3), dy.double(), padding=1).float()
        xn = to_nhwc_hp(x, ld=ldx or Ci, off=(ldx - Ci) if ldx else 0)
        dyn = to_nhwc_hp(dy)
        dwp = torch.zeros(9 * Co * Ci, dtype=torch.float32, device="cuda")
        keep.append((xn, dyn, dwp, ref, Ci, Co))
        d.f, d.f_ld, d.CF = dyn.data_ptr(), Co, Co
        d.s, d.s_ld, d.CS = xn.data_ptr() + 4 * ((ldx - Ci) if ldx else 0), ldx or Ci, Ci
        d.Hf, d.Wf, d.dw = H, W, dwp.data_ptr()
    counts = (ctypes.c_int * 8)()
    cap = lib.crimac_wgrad_group_plan(P, arr, len(specs), B, items, None, 0, counts)
    assert cap > 0, lib.crimac_last_error()
    host = torch.zeros(8 * cap * 2, dtype=torch.int32)
    assert lib.crimac_wgrad_group_plan(P, arr, len(specs), B, items, ctypes.c_void_p(host.data_ptr()), cap, counts) == cap
    dev_items = host.cuda()
    ctr = torch.zeros(8, dtype=torch.int32, device="cuda")
    call("crimac_wgrad_group", P, ctypes.byref(arr), len(specs), B, ptr(dev_items), cap, ctypes.byref(counts), ptr(ctr))
    torch.cuda.synchronize()
    for (xn, dyn, dwp, ref, Ci, Co), sp in zip(keep, specs):
        grad = torch.empty(Co, Ci, 3, 3, dtype=torch.float32, device="cuda")
        call("crimac_unpack_wgrad_conv3x3", ptr(dwp), Co, Ci, Ci, ptr(grad))
        torch.cuda.synchronize()
        assert relerr(grad.cpu(), ref) < 2 * TOL, sp
    # channel counts that are not whole 64 x 64 tiles are refused for plane pairs (they keep crimac_wgrad)
    arr[0].CS = 96
    assert lib.crimac_wgrad_group_plan(P, arr, 1, B, 0, None, 0, counts) < 0


def test_wgrad_operands_in_channel_slices():
    """F and S living in slices of wider buffers (the decoder's first convolution reads the concat buffer)."""
    B, H, W, Ci, Co = 2, 32, 32, 128, 64
    g = torch.Generator().manual_seed(7)
    x = hp_round(torch.randn(B, Ci, H, W, generator=g))
    dy = hp_round(torch.randn(B, Co, H, W, generator=g))
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 3, 3), dy.double(), padding=1).float()
    xn, dyn = to_nhwc_hp(x, ld=Ci + 64, off=64), to_nhwc_hp(dy, ld=2 * Co, off=Co)
    dwp = torch.zeros(9 * Co * Ci, dtype=torch.float32, device="cuda")
    call("crimac_wgrad", P, 0, ptr(dyn, Co), 2 * Co, Co, ptr(xn, 64), Ci + 64, Ci, B, H, W, ptr(dwp), 0)
    grad = torch.empty(Co, Ci, 3, 3, dtype=torch.float32, device="cuda")
    call("crimac_unpack_wgrad_conv3x3", ptr(dwp), Co, Ci, Ci, ptr(grad))
    torch.cuda.synchronize()
    assert relerr(grad.cpu(), ref) < 2 * TOL


@pytest.mark.parametrize("C", [64, 256])
def test_elementwise_kernels_read_fp32_and_write_plane_pairs(C):
    B, H, W = 2, 8, 8
    M = B * H * W
    g = torch.Generator().manual_seed(8)
    y = torch.randn(M, C, generator=g) * 2 + 0.5
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    yd, scd, shd = y.cuda(), sc.cuda(), sh.cuda()
    a = torch.empty(M, C, dtype=torch.float32, device="cuda")
    pool = torch.empty(M // 4, C, dtype=torch.float32, device="cuda")
    call("crimac_bn_act_pool", P, ptr(yd), C, ptr(scd), ptr(shd), 1, ptr(a), C, ptr(pool), C, B, H, W, C)
    torch.cuda.synchronize()
    ref = hp_round(torch.relu(torch.addcmul(sh, y, sc)))
    got = hp_unpack(a)
    assert relerr(got, ref) < 2e-7
    ref_pool = F.max_pool2d(got.view(B, H, W, C).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(-1, C)
    assert torch.equal(hp_unpack(pool), ref_pool)
    # inference max-pool alone: plane pairs in and out
    pool2 = torch.empty_like(pool)
    call("crimac_bn_act_pool", P, ptr(a), C, None, None, 0, None, 0, ptr(pool2), C, B, H, W, C)
    torch.cuda.synchronize()
    assert torch.equal(hp_unpack(pool2), ref_pool)
    # BatchNorm backward: fp32 da, y -> plane-pair dy
    da = torch.randn(M, C, generator=g) * 1e-2
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    act = y * sc + sh
    dz = torch.where(act > 0, da, torch.zeros_like(da)).double()
    xh = ((y - mean) * invstd).double()
    s_dz, s_dzx = dz.sum(0), (dz * xh).sum(0)
    dyr = (sc.double() * (dz - s_dz / M - xh * s_dzx / M)).float()
    dad, md, isd = da.cuda(), mean.cuda(), invstd.cuda()
    sdz, sdzx = s_dz.cuda(), s_dzx.cuda()
    dy = torch.empty(M, C, dtype=torch.float32, device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    call("crimac_bn_bwd_apply", P, ptr(dad), C, ptr(yd), C, ptr(scd), ptr(shd), ptr(md), ptr(isd), ptr(sdz), ptr(sdzx), M, M, C,
         ptr(dy), C, ptr(dg), ptr(db), None)
    torch.cuda.synchronize()
    assert relerr(hp_unpack(dy), dyr) < 2e-6


# ---- whole network -----------------------------------------------------------------------------------------------------
def _model(seed=0):
    m = pkg.UNet_Baseline(3, 4, precision="h3p")
    m.load_state_dict(synth.synth_state_dict(seed=seed))
    return m.cuda()


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def _l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def _golden_gradient_errors(m, x, lab, fix, repeats=3):
    """{parameter: (median L2-rel error vs the reference golden gradient, median |norm - golden norm| / golden norm)} over
    ``repeats`` forward + backward passes of the SAME model on the same crops.  Why a median: two runs of one precision are
    not bit-identical (statistics and column sums are added up by atomics in arrival order); a 1e-7 difference in the
    forward pass now and then puts an activation on the other side of a plane-pair rounding boundary, which moves a ReLU mask
    or a pool position -- a DISCRETE change -- and a gradient that is a sum with heavy cancellation (the transposed
    convolutions' biases) then jumps by a few 1e-3 relative: up_convs.2.upconv.bias read 3.6e-4 ... 3.1e-3 in six runs of h3p
    and once 4.28e-3 in h3f (tolerance 4.25e-3).  The median of three passes is the arithmetic's error, not one draw's tail."""
    import re
    pre_bn_bias = re.compile(r"(down_convs\.\d+\.main\.[03]|up_convs\.\d+\.conv[12])\.bias")
    crit = pkg.WeightedCrossEntropy([10.0, 300.0, 250.0]).cuda()
    errs, nerrs = {}, {}
    first = None
    for _ in range(repeats):
        for p_ in m.parameters():
            p_.grad = None
        m.train()
        logits = m(x)
        loss = crit(logits, lab.long())
        loss.backward()
        if first is None:          # (logits, loss, BatchNorm running statistics after ONE training forward)
            first = (logits.detach().clone(), float(loss),
                     {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k})
        for k, p_ in m.named_parameters():
            if pre_bn_bias.fullmatch(k):
                continue
            g_ = p_.grad.detach().cpu()
            gn = float(fix["gnorm/" + k])
            nerrs.setdefault(k, []).append(abs(float(g_.double().norm()) - gn) / gn)
            if "grad/" + k in fix.files:
                errs.setdefault(k, []).append(_l2(g_, fix["grad/" + k]))
    med = lambda v: sorted(v)[len(v) // 2]
    return {k: (med(errs[k]) if k in errs else None, med(nerrs[k])) for k in nerrs}, first


def test_network_eval_and_train_step_match_reference_golden(golden_dir):
    """The north-star bar in the fast parity precision: eval logits <= 1e-3 (here 1e-5) with IDENTICAL argmax masks vs the
    imported reference; one training step: logits, loss, BatchNorm buffers, every gradient within a few multiples of the
    reference's own fp32-vs-fp64 noise (the backward pass runs on loss-scaled fp16 plane pairs)."""
    import re
    pre_bn_bias = re.compile(r"(down_convs\.\d+\.main\.[03]|up_convs\.\d+\.conv[12])\.bias")
    fix = np.load(os.path.join(golden_dir, "full64_256.npz"))
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, 256, 256, seed=1)).cuda()
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=2)).cuda()
    m = _model().eval()
    with torch.no_grad():
        out = m(x)
    ref = torch.from_numpy(fix["logits_eval"])
    flips = int((out.argmax(1).cpu() != ref.argmax(1)).sum())
    print(f"eval h3p: rel={_rel(out, ref):.3e} argmax flips={flips}/{ref[:, 0].numel()}")
    assert _rel(out, ref) < 1e-5 and flips == 0
    errs, (logits, loss, stats) = _golden_gradient_errors(m, x, lab, fix)
    ref_t = torch.from_numpy(fix["logits_train"])
    assert _rel(logits, ref_t) < 2e-5 and int((logits.argmax(1).cpu() != ref_t.argmax(1)).sum()) == 0
    assert abs(loss - float(fix["losses"][0])) < 1e-5 * abs(float(fix["losses"][0]))
    for k, v in stats.items():
        assert _rel(v.float(), fix["stat1/" + k]) < 1e-4, k
    worst = 0.0
    for k, (r, rn) in errs.items():
        noise = float(fix["gnoise/" + k])
        tol = max(6 * noise, 3e-3)
        assert rn <= tol, (k, rn, tol)
        if r is not None:
            worst = max(worst, r / tol)
            assert r < tol, (k, r, tol)
    print("h3p worst gradient L2-rel / tolerance:", worst)
    assert m.engine.skipped_steps() == 0


# A pixel whose argmax differs from the oracle's counts as a parity failure UNLESS the oracle's own two top logits are
# closer than its fp32 round-off at that pixel (two fp32 evaluations of the reference -- another thread count, another
# oneDNN blocking -- disagree there too): the criterion is the MARGIN, never a count fitted to one seed.
TIE_MARGIN = 2e-6          # logits are O(1): 2e-6 is ~16 fp32 ulps of the oracle's own summation noise
MAX_TIE_FRACTION = 1e-5    # and such ties must stay rare (a systematic error would show as many small-margin flips)


def assert_masks_identical_up_to_oracle_ties(out, ref, what):
    diff = out.argmax(1) != ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])[diff]
    n = int(diff.sum())
    print(f"{what}: argmax flips={n}/{ref[:, 0].numel()} oracle margins at flips={[f'{v:.1e}' for v in margin.tolist()[:8]]}")
    assert bool((margin < TIE_MARGIN).all()), (what, margin.max())
    assert n <= MAX_TIE_FRACTION * ref[:, 0].numel(), (what, n)


@pytest.mark.parametrize("seed", [100, 1])
def test_network_batch32_full_size_matches_oracle(seed):
    """BASELINE configs[1] size (32 x 4 x 256 x 256) in the fast parity precision against the CPU oracle: every kernel at
    the dispatch the benchmark uses.  Identical argmax masks except where the oracle's own two top logits tie to fp32
    round-off (margin criterion above).  seed 1 = the crops bench.py's ``batch32_vs_oracle`` runs on."""
    from oracle import unet_oracle as orc
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    sd = synth.synth_state_dict(seed=0)
    x = torch.from_numpy(synth.synth_echogram_batch(32, 4, 256, 256, seed=seed))
    ref = orc.predict(sd, x)
    m = _model().eval()
    with torch.no_grad():
        out = m(x.cuda()).cpu()
    r = _rel(out, ref)
    print(f"B=32 eval h3p seed {seed}: rel={r:.3e}")
    assert r < 1e-5
    assert_masks_identical_up_to_oracle_ties(out, ref, f"B=32 eval h3p seed {seed}")


def test_training_trajectory_and_loss_scale_bookkeeping():
    """Three fused training steps (forward, weighted CE, loss-scaled backward on fp16 plane pairs, guarded SGD) track
    the reference golden trajectory; no step is skipped at the default scale."""
    fixp = os.path.join(os.path.dirname(__file__), "golden", "full64_256.npz")
    fix = np.load(fixp)
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, 256, 256, seed=1)).cuda()
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=2)).cuda()
    m = _model().train()
    eng = m.engine
    cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
    losses = [float(eng.train_step(x, lab, cw, lr=0.005, momentum=0.95)) for _ in range(3)]
    print("h3p losses", losses, "golden", fix["losses"].tolist())
    assert eng.skipped_steps() == 0
    for a, b in zip(losses, fix["losses"]):
        assert abs(a - float(b)) < 2e-3 * abs(float(b))


@pytest.mark.parametrize("cfg", [dict(in_channels=11), dict(depth=3), dict(depth=4, in_channels=6, start_filts=128),
                                 dict(batch=1, hw=(16, 48)), dict(batch=5, hw=(80, 32)),
                                 dict(depth=6, hw=(64, 96))])       # BASELINE configs[4] "deeper": 2048-channel bottleneck
def test_other_architectures_and_ragged_shapes_match_oracle(cfg):
    """Away from the benchmark shape (metadata planes as input channels, shallower / wider nets, odd batches, the smallest
    legal crop, non-square crops): eval logits and one training step against the oracle."""
    from oracle import unet_oracle as orc
    depth, cin, sf = cfg.get("depth", 5), cfg.get("in_channels", 4), cfg.get("start_filts", 64)
    B = cfg.get("batch", 2)
    H, W = cfg.get("hw", (32, 48))
    sd = synth.synth_state_dict(in_channels=cin, depth=depth, start_filts=sf, seed=9)
    x = torch.from_numpy(synth.synth_echogram_batch(B, cin, H, W, seed=91))
    lab = torch.from_numpy(synth.synth_labels(B, H, W, seed=92))
    m = pkg.UNet_Baseline(3, cin, depth=depth, start_filts=sf, precision="h3p")
    m.load_state_dict(sd)
    m.cuda().eval()
    with torch.no_grad():
        out = m(x.cuda())
    assert _rel(out, orc.predict(sd, x)) < 1e-5
    ref_loss, ref_logits, ref_grads, ref_stats = orc.loss_and_grads(sd, x, lab)
    m.train()
    crit = pkg.WeightedCrossEntropy([10.0, 300.0, 250.0]).cuda()
    logits = m(x.cuda())
    loss = crit(logits, lab.long().cuda())
    loss.backward()
    assert _rel(logits.detach(), ref_logits) < 1e-4
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    g = {k: p.grad for k, p in m.named_parameters()}
    deep = f"down_convs.{depth - 1}.main.3.weight"
    for k in ("conv_final.weight", "down_convs.0.main.0.weight", deep, "up_convs.0.upconv.weight"):
        assert _l2(g[k], ref_grads[k]) < (5e-2 if min(H, W) >> (depth - 1) <= 2 else 2e-2), k
    assert m.engine.skipped_steps() == 0


def test_tiled_inference_and_pipeline_surface_in_h3p():
    """predict_survey (gather -> plane pairs, U-Net, softmax, scatter) in the parity precision against the oracle, and the
    SegPipe surface: predict_batch + the yaml's `precision: 'h3p'`."""
    import types
    from crimac_classifiers_unet_amd import tiled_inference as ti
    from oracle import tiling_oracle as torc, unet_oracle as orc
    from tools.fake_reader import FakeZarrReader, synth_survey
    sv, labels, seabed = synth_survey()
    sv, labels, seabed = sv[:, :440, :300], labels[:440, :300], np.clip(seabed[:440], 0, 230)
    reader = FakeZarrReader(sv, labels, seabed)
    model = _model(seed=0)
    sd = synth.synth_state_dict(seed=0)
    pipe = types.SimpleNamespace(model=model, device=torch.device("cuda"), frequencies=[18, 38, 120, 200])
    chunks = list(ti.predict_survey(reader, pipe, (256, 256), 20, 2, 440))
    assert len(chunks) == 1

    def net(d):
        return orc.predict(sd, torch.from_numpy(d[None]), return_softmax=True)[0].numpy()

    ref, _ = torc.predict_chunk(sv, labels, seabed, 0, 440, net)
    out = chunks[0][2]
    assert np.array_equal(out != 0, ref != 0) and np.abs(out - ref).max() < 1e-5


@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 128), (2, 16, 16, 16, 64), (1, 32, 32, 64, 32)])
def test_reproducible_weight_gradients_in_h3p(shape):
    """crimac_wgrad_partials (one slab per pixel split by plain stores, ordered sum in crimac_unpack_wgrad_layers) with
    plane-pair operands -- the 8-wave kernel, its first-layer form and the register-staged fallback: two runs
    bit-identical, equal to the torch reference."""
    import ctypes
    B, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(41)
    x = hp_round(torch.randn(B, Ci, H, W, generator=g))
    dy = hp_round(torch.randn(B, Co, H, W, generator=g) * 30)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 3, 3), dy.double(), padding=1).float()
    dyn, xn = to_nhwc_hp(dy), to_nhwc_hp(x)
    lib = hip.load_library()
    n = 9 * Co * Ci
    outs = []
    for target in (0, 6):
        sp = lib.crimac_wgrad_splits(P, 0, Co, Ci, B, H, W, target)
        assert sp >= 1
        stride = n + 64
        for rep in range(2):
            slabs = torch.full((sp * stride,), float("nan"), dtype=torch.float32, device="cuda")
            call("crimac_wgrad_partials", P, 0, ptr(dyn), Co, Co, ptr(xn), Ci, Ci, B, H, W, ptr(slabs), stride, target)
            grad = torch.zeros(Co, Ci, 3, 3, dtype=torch.float32, device="cuda")
            d = (hip.LayerDesc * 1)()
            d[0].grad, d[0].dw, d[0].kind = grad.data_ptr(), slabs.data_ptr(), 0
            d[0].Co, d[0].Ci, d[0].Ci_pad = Co, Ci, Ci
            d[0].dw_splits, d[0].dw_stride = sp, stride
            call("crimac_unpack_wgrad_layers", ctypes.byref(d), 1)
            torch.cuda.synchronize()
            outs.append(grad.cpu())
        assert torch.equal(outs[-1], outs[-2])
        assert relerr(outs[-1], ref) < 2 * TOL


# ---- 'h3f': the plane-pair forward pass; fp16 MFMA operands in the backward pass (CRIMAC_PREC_H3F_BWD) ---------------------
def _model_f(seed=0):
    m = pkg.UNet_Baseline(3, 4, precision="h3f")
    m.load_state_dict(synth.synth_state_dict(seed=seed))
    return m.cuda()


def test_h3f_forward_is_h3p_bit_for_bit_and_gradients_meet_the_h3p_bar(golden_dir):
    """precision 'h3f': (1) eval logits are IDENTICAL to 'h3p' (bit for bit), train-mode logits, loss and BatchNorm running
    statistics equal to fp32 rounding -- the same kernels on the same operands; (2) every gradient of the golden training step is within the tolerance the
    all-plane-pair backward pass is held to (max(6 x the reference's own fp32-vs-fp64 noise, 3e-3)), norms included -- as the
    median over three passes (_golden_gradient_errors says why);
    (3) no step is skipped at the default loss scale."""
    import re
    pre_bn_bias = re.compile(r"(down_convs\.\d+\.main\.[03]|up_convs\.\d+\.conv[12])\.bias")
    fix = np.load(os.path.join(golden_dir, "full64_256.npz"))
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, 256, 256, seed=1)).cuda()
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=2)).cuda()
    mf, mp = _model_f().eval(), _model().eval()
    with torch.no_grad():
        assert torch.equal(mf(x), mp(x))
    ef, first_f = _golden_gradient_errors(mf, x, lab, fix)
    ep, first_p = _golden_gradient_errors(mp, x, lab, fix)
    # (train mode: the BatchNorm statistics are added up by atomics in arrival order, so two runs of the SAME precision
    # differ in the last bits too -- equal to fp32 rounding, not bit for bit)
    assert _rel(first_f[0], first_p[0]) < 2e-6 and abs(first_f[1] - first_p[1]) <= 1e-6 * abs(first_p[1])
    for k, v in first_p[2].items():
        assert _rel(first_f[2][k], v) < 1e-6, k
    worst, worst_p, bad = 0.0, 0.0, []
    for k, (r, rn) in ef.items():
        noise = float(fix["gnoise/" + k])
        tol = max(6 * noise, 3e-3)
        if rn > tol:
            bad.append((k, "norm", rn, tol))
        if r is not None:
            rp = ep[k][0]
            worst, worst_p = max(worst, r / tol), max(worst_p, rp / tol)
            print(f"  {k:40s} noise {noise:.2e}  h3f {r:.2e} ({r / tol:.2f} tol)  h3p {rp:.2e} ({rp / tol:.2f} tol)")
            if r >= tol:
                bad.append((k, "l2", r, rp, tol))
    print(f"worst gradient L2-rel / tolerance: h3f {worst:.3f}, h3p {worst_p:.3f}")
    assert not bad, bad
    assert mf.engine.skipped_steps() == 0


def test_h3f_training_trajectory_and_other_shapes():
    """Three fused h3f steps track the reference golden trajectory like h3p's; a ragged, deeper-than-default case and a
    wide one run against the oracle (loss + four gradients), so every shadow / personality seam sees odd shapes."""
    from oracle import unet_oracle as orc
    fix = np.load(os.path.join(os.path.dirname(__file__), "golden", "full64_256.npz"))
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, 256, 256, seed=1)).cuda()
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=2)).cuda()
    m = _model_f().train()
    eng = m.engine
    cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
    losses = [float(eng.train_step(x, lab, cw, lr=0.005, momentum=0.95)) for _ in range(3)]
    print("h3f losses", losses, "golden", fix["losses"].tolist())
    assert eng.skipped_steps() == 0
    for a, b in zip(losses, fix["losses"]):
        assert abs(a - float(b)) < 2e-3 * abs(float(b))
    # eval after training steps: the follower planes were re-packed (forward personality), logits finite
    m.eval()
    with torch.no_grad():
        assert bool(torch.isfinite(m(x)).all())
    for cfg in (dict(depth=3, in_channels=6, batch=3, hw=(48, 80)), dict(depth=4, start_filts=128, batch=1, hw=(32, 32))):
        depth, cin, sf = cfg.get("depth", 5), cfg.get("in_channels", 4), cfg.get("start_filts", 64)
        B, (H, W) = cfg["batch"], cfg["hw"]
        sd = synth.synth_state_dict(in_channels=cin, depth=depth, start_filts=sf, seed=9)
        xx = torch.from_numpy(synth.synth_echogram_batch(B, cin, H, W, seed=91))
        ll = torch.from_numpy(synth.synth_labels(B, H, W, seed=92))
        mm = pkg.UNet_Baseline(3, cin, depth=depth, start_filts=sf, precision="h3f")
        mm.load_state_dict(sd)
        mm.cuda().train()
        ref_loss, ref_logits, ref_grads, _ = orc.loss_and_grads(sd, xx, ll)
        crit = pkg.WeightedCrossEntropy([10.0, 300.0, 250.0]).cuda()
        logits = mm(xx.cuda())
        loss = crit(logits, ll.long().cuda())
        loss.backward()
        assert _rel(logits.detach(), ref_logits) < 1e-4 and abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
        g = {k: p.grad for k, p in mm.named_parameters()}
        for k in ("conv_final.weight", "down_convs.0.main.0.weight", f"down_convs.{depth - 1}.main.3.weight", "up_convs.0.upconv.weight"):
            assert _l2(g[k], ref_grads[k]) < 5e-2, (cfg, k, _l2(g[k], ref_grads[k]))
        assert mm.engine.skipped_steps() == 0


H3F = hip.PREC_H3F_BWD


@pytest.mark.parametrize("shape", [(2, 32, 32, 128, 256), (3, 20, 24, 64, 64), (2, 64, 64, 128, 64), (2, 16, 16, 256, 128)])
def test_h3f_backward_kernels_against_torch(shape):
    """The contractions and the BatchNorm-backward of CRIMAC_PREC_H3F_BWD, one by one: (1) the input gradient of a 3x3
    convolution from an fp16 dy with fp16 weight planes into an fp32 da, with the fused BatchNorm-backward sums read from an
    fp32 y -- and as an fp16 output (CRIMAC_EPI_OUT_PLANES); (2) the weight gradient with F = fp16 dy and S = the hi plane
    of a PLANE-PAIR activation (and the transposed convolution's, with the roles swapped); (3) bn_bwd_apply from fp32
    (da, y) into an fp16 dy."""
    B, H, W, Ci, Co = shape                       # forward conv Ci -> Co
    g = torch.Generator().manual_seed(51)
    M = B * H * W
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    dy = torch.randn(B, Co, H, W, generator=g).half().float()              # exact in fp16
    x = hp_round(torch.randn(B, Ci, H, W, generator=g))                     # a plane-pair activation
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl, dh, dl = (torch.empty(9 * Co * Ci, **i16), torch.empty(8, **i16), torch.empty(9 * Ci * Co, **i16), torch.empty(8, **i16))
    call("crimac_pack_conv3x3", ptr(w.cuda()), Co, Ci, Ci, None, hip.PLANES_FP16, ptr(fh), ptr(fl), ptr(dh), ptr(dl))
    w16 = w.half().float()
    dyn = dy.permute(0, 2, 3, 1).reshape(M, Co).half().cuda().contiguous()
    # (1) input gradient, fp32 out + fused sums on an fp32 y
    ref_da = torch.nn.grad.conv2d_input((B, Ci, H, W), w16.double(), dy.double(), padding=1).float()
    y_prev = torch.randn(M, Ci, generator=g) * 1.5 + 0.3
    vec = torch.stack([torch.randn(Ci, generator=g) * 0.2, torch.rand(Ci, generator=g) + 0.5,
                       torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3]).contiguous()
    R = 5
    acc = torch.zeros(2, R, Ci, dtype=torch.float64, device="cuda")
    da = torch.empty(M, Ci, dtype=torch.float32, device="cuda")
    yd, vd = y_prev.cuda(), vec.cuda()
    call("crimac_conv3x3", H3F, ptr(dyn), Co, B, H, W, Co, Ci, ptr(dh), ptr(dl), None, ptr(da), Ci, 0, 2, ptr(acc[0]),
         ptr(acc[1]), R, ptr(yd), Ci, ptr(vd), Ci)
    torch.cuda.synchronize()
    assert relerr(from_nhwc(da, B, H, W), ref_da) < 5e-6
    dz = torch.where((y_prev * vec[2] + vec[3]) > 0, da.cpu(), torch.zeros(()))
    xhat = (y_prev - vec[0]) * vec[1]
    assert relerr(acc[0].sum(0).cpu(), dz.double().sum(0)) < 1e-5 and relerr(acc[1].sum(0).cpu(), (dz * xhat).double().sum(0)) < 1e-5
    # ... and with an fp16 output (the up half of a decoder block's d(concat))
    da16 = torch.empty(M, Ci, dtype=torch.float16, device="cuda")
    call("crimac_conv3x3", H3F, ptr(dyn), Co, B, H, W, Co, Ci, ptr(dh), ptr(dl), None, ptr(da16), Ci, hip.EPI_OUT_PLANES, 0,
         None, None, 1, None, 0, None, 0)
    torch.cuda.synchronize()
    assert relerr(from_nhwc(da16, B, H, W), ref_da) < 1e-3
    # (2) weight gradient: F = fp16 dy, S = hi plane of the plane-pair activation
    xn = to_nhwc_hp(x)
    x_hi = x.half().float()
    ref_dw = torch.nn.grad.conv2d_weight(x_hi.double(), (Co, Ci, 3, 3), dy.double(), padding=1).float()
    dwp = torch.zeros(9 * Co * Ci, dtype=torch.float32, device="cuda")
    call("crimac_wgrad", H3F, 0, ptr(dyn), Co, Co, ptr(xn), Ci, Ci, B, H, W, ptr(dwp), 0)
    grad = torch.empty(Co, Ci, 3, 3, dtype=torch.float32, device="cuda")
    call("crimac_unpack_wgrad_conv3x3", ptr(dwp), Co, Ci, Ci, ptr(grad))
    torch.cuda.synchronize()
    assert relerr(grad.cpu(), ref_dw) < 2e-5
    # transposed convolution Ci -> Co on the coarse grid: F = plane-pair x (coarse), S = fp16 dY (fine)
    dyf = torch.randn(B, Co, 2 * H, 2 * W, generator=g).half().float()
    wg = torch.zeros(Ci, Co, 2, 2, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(x_hi.double(), wg, None, stride=2).backward(dyf.double())
    dyfn = dyf.permute(0, 2, 3, 1).reshape(4 * M, Co).half().cuda().contiguous()
    dwu = torch.zeros(4 * Ci * Co, dtype=torch.float32, device="cuda")
    call("crimac_wgrad", H3F, 1, ptr(xn), Ci, Ci, ptr(dyfn), Co, Co, B, H, W, ptr(dwu), 0)
    gu = torch.empty(Ci, Co, 2, 2, dtype=torch.float32, device="cuda")
    call("crimac_unpack_wgrad_upconv2x2", ptr(dwu), Ci, Co, ptr(gu))
    torch.cuda.synchronize()
    assert relerr(gu.cpu(), wg.grad.float()) < 2e-5
    # (3) bn_bwd_apply: (da fp32, y fp32) -> dy fp16, against the plane-pair form of the same kernel
    sums = torch.zeros(2, R, Ci, dtype=torch.float64, device="cuda")
    sums[0, 0], sums[1, 0] = acc[0].sum(0), acc[1].sum(0)
    out16 = torch.empty(M, Ci, dtype=torch.float16, device="cuda")
    outpp = torch.empty(M, Ci, dtype=torch.float32, device="cuda")
    dgam, dbet = torch.empty(Ci, device="cuda"), torch.empty(Ci, device="cuda")
    for prec_, o in ((H3F, out16), (P, outpp)):
        call("crimac_bn_bwd_apply_replicas", prec_, ptr(da), Ci, ptr(yd), Ci, ptr(vd), Ci, ptr(sums[0]), ptr(sums[1]), R, M, M, Ci,
             ptr(o), Ci, ptr(dgam), ptr(dbet))
    torch.cuda.synchronize()
    want = hp_unpack(outpp)
    assert relerr(out16.float().cpu(), want) < 1e-3
    assert float((out16.float().cpu() - want).abs().max()) <= 2.0 ** -10 * float(want.abs().max())
