"""Execution plan of the U-Net hot path on one MI355X.

Drives the C-ABI kernels (include/crimac_unet_hip.h) for ``UNet_Baseline.forward``
(reference crimac_unet/models/unet.py:327-343), its backward (``loss.backward()``,
pipeline.py:177), the weighted cross entropy (pipeline.py:132-141) and the SGD-momentum update
(pipeline.py:156, :178).

Data layout in HBM
  * activations NHWC, bf16 (``precision='bf16'``) or fp32 (``'f32x3'``); ``torch.cat((up, skip), 1)``
    (unet.py:132) is never executed: the transposed conv and the encoder write into the two channel
    halves of one ``[M, 2C]`` buffer (pixel stride ``ld = 2C``);
  * parameters fp32 in ONE flat buffer (module parameters are views into it), gradients and momentum
    likewise: one SGD launch, a few large all-reduce buckets;
  * MFMA weight operands: bf16 hi(+lo) planes ``[tap][Cout][Cin]`` (forward) and ``[tap][Cin][Cout]``
    (input gradient), re-packed from the fp32 masters after every update.
PyTorch supplies memory and streams only; there is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import hip
from .hip import Act, call, ptr


def _on_device(fn):
    """Run an engine entry point with the engine's GPU as the current device: the kernels are launched on
    ``torch.cuda.current_stream()``, which is per device (a model on cuda:1 used while cuda:0 is current would
    otherwise be launched on the wrong GPU's stream)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        if getattr(self, "device", None) is None:
            self.bind()                               # (first use: learn which GPU the parameters live on)
        dev = self.device
        if dev is None or dev.type != "cuda":
            return fn(self, *a, **kw)
        with torch.cuda.device(dev):
            return fn(self, *a, **kw)
    return wrapped

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
CIN_PAD = 16          # first-layer input channels are zero-padded to one MFMA k-step
STAT_REPLICAS = 64    # conv epilogues spread their BatchNorm partial sums over this many accumulators


def _align(n, a=64):
    return (n + a - 1) // a * a


class _ConvBlock:
    """conv3x3 + BatchNorm2d (+ReLU): names of its tensors and its packed operands."""

    def __init__(self, conv_key, bn_key, cin, cout, cin_pad=None):
        self.conv_key, self.bn_key = conv_key, bn_key
        self.cin, self.cout = cin, cout
        self.cin_pad = cin_pad or cin
        self.idx = None            # index among BN layers (stat scratch slot)


class _UpConv:
    def __init__(self, key, cin, cout):
        self.key, self.cin, self.cout = key, cin, cout


class UNetEngine:
    def __init__(self, module, precision="bf16", leader=None):
        """``leader``: another engine of the SAME module that owns the flat parameter / gradient / momentum buffers.
        A follower engine runs eval-mode forwards in its own precision on the leader's parameters (its own packed
        operands and activation buffers): the inference precision of a model that trains in another one
        (``UNet_Baseline(infer_precision=...)``)."""
        if precision not in hip.PREC_NAMES:
            raise ValueError(f"precision must be one of {list(hip.PREC_NAMES)}, got {precision!r}")
        hip.load_library()       # fail loudly before anything else if the HIP library is missing
        self.module = module
        self.leader = leader
        self._followers = []
        if leader is not None:
            leader._followers.append(self)
        self.precision = precision
        self.prec = hip.PREC_NAMES[precision]
        self.act_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(precision, torch.float32)
        self.planes = hip.PREC_PLANES[self.prec]
        self.planes_arg = hip.PREC_PLANES_ARG[self.prec]     # what the packing entry points take
        self.is16 = self.prec in hip.PREC_16BIT
        # h3p: MFMA operand tensors (activations, output gradients) are fp16 plane pairs split by their producer; the
        # buffers have fp32 addressing (4 bytes per element), so fp32 and plane-pair tensors share concat buffers
        self.is_hp = self.prec == hip.PREC_H3P
        self.lds_dma = self.is16 or self.is_hp           # the LDS-DMA kernel family (and its fused epilogues) applies
        self.prec_bwd = hip.PREC_BACKWARD.get(self.prec, self.prec)      # (f32h3: backward on the bf16 2-plane split)
        # fp16 storage: the loss gradient is scaled by `loss_scale` (crimac_wce_bwd upstream) so that activation
        # gradients (1e-5 .. 1e-8 unscaled) sit in fp16's normal range; SGD divides it out again and skips the
        # step when a gradient overflowed (crimac_grad_overflow_flag / crimac_sgd_momentum_guarded)
        # (h3p: the output gradients dy are fp16 PLANE PAIRS -- same range, 11 more bits -- and travel scaled likewise)
        self.loss_scale = 2.0 ** 16 if precision in ("fp16", "h3p", "h3f") else 1.0
        self.dynamic_loss_scale = precision in ("fp16", "h3p", "h3f")
        # h3f: forward = h3p (plane pairs, 3 MFMAs per product: the logits, the loss and the BatchNorm statistics are
        # h3p's); in the backward pass only the MFMA OPERANDS drop to plain fp16 (1 MFMA per product): the output gradients
        # dy are stored as fp16, the input-gradient weight planes are fp16, and the weight gradients read the hi plane of
        # the plane-pair activations.  Everything the backward pass DECIDES on -- ReLU masks, xhat, max-pool positions --
        # still comes from the forward pass's fp32 y, and activation gradients da stay fp32 (CRIMAC_PREC_H3F_BWD).  A
        # first form that ran the whole fp16 backward pass on fp16 copies of y was 4e-2 off on the first layer's gradient:
        # rounding y flips ReLU masks and pool positions (discrete errors), rounding a gradient does not.
        self.bwd16 = precision == "h3f"
        env = os.environ.get("CRIMAC_FUSE_UNPOOL_APPLY")
        self.fuse_unpool_apply = (env != "0") if env is not None else precision not in ("bf16", "fp16")
        if self.bwd16:
            self.prec_bwd = hip.PREC_H3F_BWD
        self._scale_state = None            # int32[2] on the GPU: [overflow this step, steps skipped]
        self._skipped_seen = 0
        self._good_checks = 0
        self.depth = module.depth
        self.sf = module.start_filts
        self.in_channels = module.in_channels
        self.n_classes = module.n_classes
        if self.sf % 64 != 0:
            raise ValueError("the MFMA tiles need start_filts to be a multiple of 64 "
                             f"(got {self.sf}); the reference pipeline uses 64 (pipeline.py:394)")
        if self.in_channels > CIN_PAD:
            raise ValueError(f"in_channels={self.in_channels} > {CIN_PAD} is not supported")
        if not 2 <= self.n_classes <= 4:
            raise ValueError("n_classes must be in 2..4")
        D, sf = self.depth, self.sf
        self.enc, self.dec, self.ups = [], [], []
        for i in range(D):
            c = sf * 2 ** i
            cin = self.in_channels if i == 0 else c // 2
            p = f"down_convs.{i}.main."
            self.enc.append((_ConvBlock(p + "0", p + "1", cin, c, CIN_PAD if i == 0 else None),
                             _ConvBlock(p + "3", p + "4", c, c)))
        for j in range(D - 1):
            cprev = sf * 2 ** (D - 1 - j)
            c = cprev // 2
            p = f"up_convs.{j}."
            self.ups.append(_UpConv(p + "upconv", cprev, c))
            self.dec.append((_ConvBlock(p + "conv1", p + "bn1", 2 * c, c),
                             _ConvBlock(p + "conv2", p + "bn2", c, c)))
        # UNet_LateMetInject (unet.py:346-391): per-pixel perceptron on the metadata planes, its output concatenated
        # to the last decoder activation in front of conv_final (65 -> n_classes)
        self.lmi = bool(getattr(module, "late_meta_inject", False))
        self.meta_channels = int(getattr(module, "meta_in_channels", 0)) if self.lmi else 0
        if self.lmi and not 1 <= self.meta_channels <= 8:
            raise ValueError(f"late metadata injection supports 1..8 metadata channels, got {self.meta_channels}")
        if self.lmi and self.bwd16:
            raise NotImplementedError("precision 'h3f' with late metadata injection (use 'h3p')")
        if self.bwd16:
            # h3f's fp16 backward pass exists on the fused kernels only: checked HERE, not in the first backward pass
            # (ADVICE r4): every transposed convolution inside crimac_upconv2x2_dgrad_bnb_prec's shapes, the fused
            # BatchNorm-backward paths switched on
            bad = [u.key for u in self.ups if u.cout % 64 != 0 or u.cin % 128 != 0]
            if bad:
                raise NotImplementedError(f"precision 'h3f': transposed convolutions {bad} lie outside the fused input-gradient "
                                          "kernel (Cout % 64, Cin % 128); precision 'h3p' covers them")
            off = [n for n, v in (("CRIMAC_FUSE_BNB", self.fuse_bn_bwd), ("CRIMAC_FUSE_UPBNB", self.fuse_up_bnb)) if not v]
            if off or not self.lds_dma:
                raise NotImplementedError("precision 'h3f' needs the fused BatchNorm-backward kernels"
                                          + (f" ({', '.join(off)} = 0 switches them off)" if off else "")
                                          + "; precision 'h3p' runs without them")
        self.blocks = [b for pair in self.enc for b in pair] + [b for pair in self.dec for b in pair]
        for k, b in enumerate(self.blocks):
            b.idx = k
        self.device = None
        self._bufs = {}
        self._train_pack_dirty = True
        self._eval_pack_dirty = True
        self._versions = None
        self.saved = None          # activations of the last train-mode forward
        self.forward_generation = 0   # bumped by every train-mode forward (unet._UNetFunction checks it in backward)
        self.accumulate_grads = False   # True: backward() adds to the flat gradient (torch semantics
        #                                 when .grad is not zeroed between backward calls)

    # ------------------------------------------------------------------------------------------
    # parameter storage
    # ------------------------------------------------------------------------------------------
    def bind(self):
        """(Re)point every module parameter at a slice of the flat fp32 buffer on the module's GPU."""
        if self.leader is not None:
            L = self.leader
            L.bind()
            if getattr(self, "flat_p", None) is not L.flat_p:          # (first use, or the leader re-bound: new device)
                self.device, self.layout, self.n_flat = L.device, L.layout, L.n_flat
                self.flat_p, self.flat_g, self.flat_v = L.flat_p, L.flat_g, L.flat_v
                self.P, self.G, self.Bf = L.P, L.G, L.Bf
                self._alloc_static()
                self._train_pack_dirty = self._eval_pack_dirty = True
                self._bufs = {}
            return
        params = list(self.module.named_parameters())
        dev = params[0][1].device
        if dev.type != "cuda":
            raise hip.HipLibraryError(
                "UNet_Baseline (MI355X build) must live on a GPU: call .to('cuda'); "
                "there is no CPU fallback for the hot path")
        ok = self.device == dev and getattr(self, "flat_p", None) is not None
        if ok:
            base = self.flat_p.data_ptr()
            for name, p in params:
                off, n, _ = self.layout[name]
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
        if ok:
            return
        self.device = dev
        self.layout, off = {}, 0
        for name, p in params:
            self.layout[name] = (off, p.numel(), tuple(p.shape))
            off += _align(p.numel())
        self.n_flat = off
        flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name, p in params:
                o, n, shp = self.layout[name]
                flat_p[o:o + n].copy_(p.detach().reshape(-1).float())
                p.data = flat_p[o:o + n].view(shp)
                p.grad = self.flat_g[o:o + n].view(shp)
        self.flat_p = flat_p
        self.P = {name: p for name, p in params}
        self.G = {name: self.flat_g[o:o + n].view(shp) for name, (o, n, shp) in self.layout.items()}
        self.Bf = dict(self.module.named_buffers())
        self._alloc_static()
        self._train_pack_dirty = self._eval_pack_dirty = True
        self._bufs = {}

    def _alloc_static(self):
        dev = self.device
        i16 = torch.int16
        self._ltab = None
        self.pk = self.pk_main = {}      # train-mode operand planes (pk_main: the forward personality's, whatever self.pk is)
        self.pk_eval = {}  # eval-mode (BN folded) forward planes + folded bias
        self.dw_plan = None       # (B, H, W) the partial-sum workspace of the weight gradients is laid out for
        self.dw_off = {}          # layer key -> (first float, slabs, floats per slab) in self.dw_packed
        self.dw_packed = None
        # interleaved plane pairs (h3p): the `hi` buffer holds both planes, `lo` is a placeholder
        il = self.is_hp
        m_hi = 2 if il else 1
        for b in self.blocks:
            n_f = 9 * b.cout * b.cin_pad
            n_lo = max(self.planes - 1, 1)        # the lo buffer holds planes 1..planes-1
            has_dg = b.cin_pad == b.cin
            self.pk[b.conv_key] = {
                "fwd_hi": torch.empty(m_hi * n_f, dtype=i16, device=dev),
                "fwd_lo": torch.empty(8 if il else n_lo * n_f, dtype=i16, device=dev),
                "dg_hi": torch.empty(m_hi * 9 * b.cin * b.cout, dtype=i16, device=dev) if has_dg else None,
                "dg_lo": torch.empty(8 if il else n_lo * 9 * b.cin * b.cout, dtype=i16, device=dev) if has_dg else None,
            }
            # fragment-major planes (round 5, CRIMAC_EPI_WFRAG): the 16-bit and plane-pair modes pack the planes the
            # channel-split kernel reads as WHOLE 128-channel ranges so that its weight loads are whole cache lines.  Forward
            # plane: N = cout in multiples of 128 and whole 64-deep chunks (of halves: 32 channels of a plane pair);
            # input-gradient plane: N = cin likewise -- except a decoder block's conv1 whose two d(concat) halves (cin / 2
            # channels each) are launched separately and are not multiples of 128
            wf = self.wfrag and (self.is16 or self.is_hp) and self.conv_impl == "halo"
            kq = 32 if self.is_hp else 64
            split_ok = not (b.cin == 2 * b.cout and (b.cin // 2) % 128 != 0)
            self.pk[b.conv_key]["fwd_frag"] = bool(wf and b.cout % 128 == 0 and b.cin_pad % kq == 0)
            self.pk[b.conv_key]["dg_frag"] = bool(wf and has_dg and b.cin % 128 == 0 and b.cout % kq == 0 and split_ok)
            # rows form of the channel-split kernel (CRIMAC_EPI_WROWS, 16-bit storage): 64-channel tiles, fragment-major planes
            fr, dr = self._rows_choice(b, has_dg) if (wf and self.is16) else (False, False)
            self.pk[b.conv_key]["fwd_rows"], self.pk[b.conv_key]["dg_rows"] = fr, dr
            self.pk[b.conv_key]["fwd_frag"] |= fr
            self.pk[b.conv_key]["dg_frag"] |= dr
            self.pk[b.conv_key]["frag_ok"] = (self.pk[b.conv_key]["fwd_frag"], self.pk[b.conv_key]["dg_frag"])
            self.pk_eval[b.conv_key] = {
                "fwd_hi": torch.empty(m_hi * n_f, dtype=i16, device=dev),
                "fwd_lo": torch.empty(8 if il else n_lo * n_f, dtype=i16, device=dev),
                "bias": torch.empty(b.cout, dtype=torch.float32, device=dev),
                "fwd_frag": self.pk[b.conv_key]["fwd_frag"], "fwd_rows": fr,
            }
        for u in self.ups:
            n = 4 * u.cin * u.cout
            n_lo = max(self.planes - 1, 1)
            self.pk[u.key] = {"fwd_hi": torch.empty(m_hi * n, dtype=i16, device=dev),
                              "fwd_lo": torch.empty(8 if il else n_lo * n, dtype=i16, device=dev),
                              "dg_hi": torch.empty(m_hi * n, dtype=i16, device=dev),
                              "dg_lo": torch.empty(8 if il else n_lo * n, dtype=i16, device=dev)}
        self.pk16 = {}     # h3f: the fp16 personality's operand planes (input-gradient planes are what it uses)
        self._ltab16 = None
        if self.bwd16:
            for b in self.blocks:
                n_f = 9 * b.cout * b.cin_pad
                has_dg = b.cin_pad == b.cin
                self.pk16[b.conv_key] = {
                    "fwd_hi": torch.empty(n_f, dtype=i16, device=dev), "fwd_lo": torch.empty(8, dtype=i16, device=dev),
                    "dg_hi": torch.empty(9 * b.cin * b.cout, dtype=i16, device=dev) if has_dg else None,
                    "dg_lo": torch.empty(8, dtype=i16, device=dev) if has_dg else None,
                    # (the fp16 personality's input-gradient planes: the 16-bit rule)
                    "fwd_frag": False,
                    "dg_frag": bool(self.wfrag and self.conv_impl == "halo" and has_dg and b.cin % 128 == 0 and b.cout % 64 == 0
                                    and not (b.cin == 2 * b.cout and (b.cin // 2) % 128 != 0))}
                dr = self._rows_choice(b, has_dg)[1] if (self.wfrag and self.conv_impl == "halo") else False
                self.pk16[b.conv_key]["dg_rows"] = dr
                self.pk16[b.conv_key]["dg_frag"] |= dr
                self.pk16[b.conv_key]["frag_ok"] = (False, self.pk16[b.conv_key]["dg_frag"])
            for u in self.ups:
                n = 4 * u.cin * u.cout
                self.pk16[u.key] = {"fwd_hi": torch.empty(n, dtype=i16, device=dev), "fwd_lo": torch.empty(8, dtype=i16, device=dev),
                                    "dg_hi": torch.empty(n, dtype=i16, device=dev), "dg_lo": torch.empty(8, dtype=i16, device=dev)}
        cmax = self.sf * 2 ** (self.depth - 1)
        nb = len(self.blocks)
        # fp64 scratch: [0:2] loss sums, then per BN layer (2*R + 2) x cmax:
        # sum[R][C], sumsq[R][C] (R replicas filled by the conv epilogues), sum_dz[C], sum_dz_xhat[C]
        self.stat = torch.zeros(2 + nb * (2 * STAT_REPLICAS + 2) * cmax, dtype=torch.float64, device=dev)
        self.cmax = cmax
        # transposed-conv bias gradients come out of the dgrad epilogue of the block above: [level][2][R][2*cmax]
        # (+ 128 doubles at the end: the 8 queue counters of up to 32 grouped weight-gradient launches per backward pass,
        # uint32 -- zeroed by the same fill)
        self.bias_scr = torch.zeros(max(self.depth - 1, 1) * 2 * STAT_REPLICAS * 2 * cmax + 128, dtype=torch.float64,
                                    device=dev)
        self._wg_counters = self.bias_scr[-128:].view(torch.int32)
        self._wg_plans = {}
        # fp32 per BN layer: mean, invstd, scale, shift
        self.bnf = torch.zeros(nb * 4 * cmax, dtype=torch.float32, device=dev)
        self.class_w = None

    def _stat(self, b, k):
        """k: 0 sum[R][C], 1 sumsq[R][C], 2 sum_dz[C], 3 sum_dz_xhat[C] of BN layer b."""
        R = STAT_REPLICAS
        base = 2 + b.idx * (2 * R + 2) * self.cmax
        if k < 2:
            o = base + k * R * self.cmax
            return self.stat[o:o + R * b.cout]
        o = base + (2 * R + k - 2) * self.cmax
        return self.stat[o:o + b.cout]

    def _bnf(self, b, k):
        o = (b.idx * 4 + k) * self.cmax
        return self.bnf[o:o + b.cout]

    # BatchNorm statistics are finished INSIDE the kernel that applies them (crimac_bn_train_act_pool,
    # crimac_bn_bwd_apply_replicas): no bn_finalize / sum_replicas launch between a convolution and its consumer (36
    # launches of a step and their dependency bubbles).  Every consumer workgroup then reads replicas x C fp64 pairs
    # from L2, so wide layers use fewer replicas (they also have fewer producer workgroups per address): replicas x C <= 4096 is
    # 16 pairs per thread, one batch of loads.
    fold_bn_finalize = os.environ.get("CRIMAC_FOLD_BNFIN", "1") != "0"

    def _nrep(self, c):
        """Replica accumulators a producer spreads the per-channel sums of a c-channel BatchNorm layer over."""
        if not self.fold_bn_finalize:
            return STAT_REPLICAS
        r = STAT_REPLICAS
        while r > 4 and r * c > 4096:
            r //= 2
        return r

    # Reproducible weight gradients (opt-in): crimac_wgrad_partials writes one slab per pixel split by plain stores
    # (no atomics, no zero fill) and crimac_unpack_wgrad_layers adds the slabs up in a fixed order, so the 31 M
    # weight gradients are bit-identical run to run.  Measured at B = 32 (profiles/r02_wgrad_partials_ab.txt): the
    # wgrad launches get 10 % faster (157 -> 142 us: the atomic tail is gone), but the step gets 3 % SLOWER (13.04 ->
    # 13.43 ms): 1.44 GB of slabs per step (75 MB per layer whatever its size: 512 workgroups x 147 KB) have to be
    # written and read back, against atomics that overlap with the other workgroups' MFMAs -- hence not the default.
    # The number of splits depends on the batch geometry, so the workspace is planned per (B, H, W).
    use_wgrad_partials = os.environ.get("CRIMAC_WGRAD_PARTIALS", "0") != "0"

    def _plan_dw(self, B, H, W):
        if self.dw_plan == (B, H, W, self.use_wgrad_partials):
            return
        lib = hip.load_library()
        geo = self._geom(B, H, W)
        D = self.depth
        off, tot = {}, 0

        def add(key, mode, cf, cs, n, h, w):
            nonlocal tot
            sp = 1
            if self.use_wgrad_partials:
                sp = lib.crimac_wgrad_splits(self.prec_bwd, mode, cf, cs, B, h, w, self.wgrad_target_blocks)
                if sp < 1:
                    raise hip.HipLibraryError(f"crimac_wgrad_splits failed for {key}")
            stride = _align(n)
            off[key] = (tot, sp, stride)
            tot += sp * stride

        for i in range(D):
            h, w, _ = geo[i]
            for b in self.enc[i]:
                add(b.conv_key, 0, b.cout, b.cin_pad, 9 * b.cout * b.cin_pad, h, w)
        for j in range(D - 1):
            L = D - 2 - j
            h, w, _ = geo[L]
            for b in self.dec[j]:
                add(b.conv_key, 0, b.cout, b.cin_pad, 9 * b.cout * b.cin_pad, h, w)
            u = self.ups[j]
            hp, wp, _ = geo[L + 1]
            add(u.key, 1, u.cin, u.cout, 4 * u.cin * u.cout, hp, wp)
        self.dw_off = off
        self.dw_packed = torch.zeros(tot, dtype=torch.float32, device=self.device)
        self.dw_plan = (B, H, W, self.use_wgrad_partials)
        self._ltab = None                             # the layer table carries the slab pointers

    def _dw(self, key):
        """(slab workspace of the layer, slabs, floats per slab)"""
        o, sp, stride = self.dw_off[key]
        return self.dw_packed[o:o + sp * stride], sp, stride

    def mark_dirty(self):
        self._train_pack_dirty = self._eval_pack_dirty = True
        self._packed_groups = set()       # layer groups whose train-mode planes were re-packed after this change
        for f in self._followers:         # (engines of other precisions on the same parameters)
            f._train_pack_dirty = f._eval_pack_dirty = True
            f._packed_groups = set()

    def _check_versions(self):
        """Detect in-place edits made through torch (load_state_dict, manual init, optimizers)."""
        v = sum(p._version for p in self.P.values()) + sum(b._version for b in self.Bf.values())
        if v != self._versions:
            self._versions = v
            self.mark_dirty()

    # ------------------------------------------------------------------------------------------
    # operand packing
    # ------------------------------------------------------------------------------------------
    def _layer_table(self):
        """crimac_layer_desc array of every conv / transposed-conv layer, ordered by the group in which
        the backward pass completes them (grad_ranges): decoder, bottleneck block, other encoder blocks."""
        if getattr(self, "_ltab", None) is not None:
            return self._ltab
        groups = [[] for _ in range(1 + self.n_enc_groups)]
        for j in range(self.depth - 1):
            groups[0] += [self.ups[j], *self.dec[j]]
        for i in range(self.depth):
            groups[self._enc_group(i)] += list(self.enc[i])
        layers = [l for g in groups for l in g]
        arr = (hip.LayerDesc * len(layers))()
        for d, l in zip(arr, layers):
            up = isinstance(l, _UpConv)
            key = l.key if up else l.conv_key
            pk = self.pk_main[key]                    # (never the fp16 personality's planes of an h3f backward pass)
            cin_pad = l.cin if up else l.cin_pad
            n = (4 if up else 9) * l.cout * cin_pad
            d.w = self.P[key + ".weight"].data_ptr()
            d.grad = self.G[key + ".weight"].data_ptr()
            if self.dw_off:
                dwt, sp, stride = self._dw(key)
                d.dw, d.dw_splits, d.dw_stride = dwt.data_ptr(), sp, stride
            else:                                     # (packing before the first backward pass: no gradients yet)
                d.dw, d.dw_splits, d.dw_stride = None, 1, 0
            d.fwd_hi, d.fwd_lo = pk["fwd_hi"].data_ptr(), pk["fwd_lo"].data_ptr()
            d.dg_hi = pk["dg_hi"].data_ptr() if pk["dg_hi"] is not None else None
            d.dg_lo = pk["dg_lo"].data_ptr() if pk["dg_lo"] is not None else None
            d.kind, d.Co, d.Ci, d.Ci_pad = (1 if up else 0), l.cout, l.cin, cin_pad
            if not up:
                d.kind |= (hip.LAYER_FWD_FRAG if pk.get("fwd_frag") else 0) | (hip.LAYER_DG_FRAG if pk.get("dg_frag") else 0)
        bounds, pos = [], 0
        for g in groups:
            bounds.append((pos, len(g)))
            pos += len(g)
        self._ltab = (arr, bounds)
        if self.bwd16:                                # the same layers with the fp16 personality's planes
            arr16 = (hip.LayerDesc * len(layers))()
            for d16, d, l in zip(arr16, arr, layers):
                C.memmove(C.byref(d16), C.byref(d), C.sizeof(hip.LayerDesc))
                pk = self.pk16[l.key if isinstance(l, _UpConv) else l.conv_key]
                d16.fwd_hi, d16.fwd_lo = pk["fwd_hi"].data_ptr(), pk["fwd_lo"].data_ptr()
                d16.dg_hi = pk["dg_hi"].data_ptr() if pk["dg_hi"] is not None else None
                d16.dg_lo = pk["dg_lo"].data_ptr() if pk["dg_lo"] is not None else None
                d16.kind = (d.kind & 1) | (hip.LAYER_DG_FRAG if pk.get("dg_frag") else 0)
            self._ltab16 = arr16
        return self._ltab

    def _pack_group(self, gi):
        arr, bounds = self._layer_table()
        first, n = bounds[gi]
        if n:
            call("crimac_pack_layers", C.byref(arr, first * C.sizeof(hip.LayerDesc)), n, self._fwd_planes_arg())
            if self.bwd16:
                call("crimac_pack_layers", C.byref(self._ltab16, first * C.sizeof(hip.LayerDesc)), n, hip.PLANES_FP16)

    def _pk_bwd(self, key):
        """Operand planes the backward pass's contractions read (h3f: the fp16 input-gradient planes)."""
        return self.pk16[key] if self.bwd16 else self.pk[key]

    def _fwd_planes_arg(self):
        """`planes` of the forward personality (inside an h3f backward pass self.planes_arg is the fp16 personality's)."""
        return hip.PREC_PLANES_ARG[hip.PREC_NAMES[self.precision]]

    def _pack_train(self):
        if not self._train_pack_dirty:
            return
        arr, bounds = self._layer_table()
        done = getattr(self, "_packed_groups", set())
        if done:                                      # (groups re-packed right behind their optimiser step)
            for gi in range(len(bounds)):
                if gi not in done:
                    self._pack_group(gi)
        else:
            call("crimac_pack_layers", C.byref(arr), len(arr), self._fwd_planes_arg())
            if self.bwd16:
                call("crimac_pack_layers", C.byref(self._ltab16), len(arr), hip.PLANES_FP16)
        self._packed_groups = set()
        self._train_pack_dirty = False

    # Weight gradients on a second stream: wgrad(layer) and the input-gradient convolution of the same layer
    # both only read dz, so they can share the GPU -- the ragged last round of one kernel and the launch gap
    # behind it are filled by the other.  The packed gradients are joined back before they are unpacked.
    # (CRIMAC_WGRAD_STREAM = number of side streams, round-robin; 0 = everything on the caller's stream)
    wgrad_side_streams = int(os.environ.get("CRIMAC_WGRAD_STREAM", "1"))
    unpack_on_side = os.environ.get("CRIMAC_UNPACK_SIDE", "1") != "0"
    early_sgd = os.environ.get("CRIMAC_EARLY_SGD", "1") != "0"
    # N > 1: apply SGD / re-pack per gradient range behind GradSync.finish_range instead of one optimiser step behind
    # finish().  OPT-IN until it has run on a multi-GPU node: its RCCL branches (stream-ordered waits, chained
    # reduce-scatter + all-gather, per-range re-pack) are covered by a single-rank nccl test and a 2-rank test that
    # needs two GPUs (tests/test_gpu_unet.py), not yet by hardware with N > 1
    early_sgd_multi = os.environ.get("CRIMAC_EARLY_SGD_MULTI", "0") != "0"
    # bench.py (N > 1): list that receives (event after the backward pass, event after the last collective was waited
    # for) per step -- the part of the gradient exchange the step could not hide
    exchange_probe = None
    early_pack = os.environ.get("CRIMAC_EARLY_PACK", "1") != "0"
    split_skip_dgrad = os.environ.get("CRIMAC_SPLIT_SKIP", "1") != "0"    # skip half of decoder dgrads on the side stream
    # launch the gradient collectives from the side stream too (they then never hold up the caller's stream)
    exchange_on_side = os.environ.get("CRIMAC_EXCHANGE_SIDE", "0") != "0"
    _side = None
    _side_events = None
    _eval_side = None

    @staticmethod
    def _gloo_ranks():
        """More than one rank over gloo (CPU-side rehearsals of the multi-GPU path): gloo's CUDA collectives
        synchronise with the device in a way that stalls for ~1 s per step once a second stream carries work
        (measured: 32 ms -> 870-1900 ms per step, two ranks on one GPU); RCCL (backend "nccl") orders itself with
        stream events and is what multi-GPU runs use."""
        import torch.distributed as dist
        return (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                and dist.get_backend() == "gloo")

    # Grouped weight gradients (crimac_wgrad_group): the conv3x3 layers of a backward group are contracted by ONE
    # persistent launch instead of one launch each -- their atomic flushes then drain under each other's MFMAs instead of
    # ending every launch (0.44 ms of the 12.0 ms bf16 step in the no-flush ablation).  The layers are collected while the
    # backward pass walks them (their dy buffers stay alive: one per block) and launched where the group's gradient range
    # is handed on (_unpack_group), or earlier once `wgrad_group_layers` of them are waiting.
    # CRIMAC_WGRAD_GROUP=0: one launch per layer (crimac_wgrad), as before.
    # Plane pairs (h3p): measured SLOWER grouped (step 25.34 ms with one launch per layer; 25.6 grouped with one item per
    # workgroup, 25.8-26.1 with persistent workgroups) although the launches themselves get faster alone (6.54 -> 6.3 ms
    # serialized): the long plane-pair items leave the input-gradient chain on the caller's stream waiting for CUs.  Off
    # for h3p unless CRIMAC_WGRAD_GROUP_H3P=1.
    wgrad_group = os.environ.get("CRIMAC_WGRAD_GROUP", "1") != "0"
    wgrad_group_h3p = os.environ.get("CRIMAC_WGRAD_GROUP_H3P", "0") != "0"
    wgrad_group_layers = int(os.environ.get("CRIMAC_WGRAD_GROUP_LAYERS", str(hip.WGRAD_GROUP_MAX_LAYERS)))
    wgrad_group_items = int(os.environ.get("CRIMAC_WGRAD_GROUP_ITEMS", "0"))      # items per layer (0: the library's default)
    _wg_pending = None

    def _groupable(self, prec, mode, cf, cs):
        if not (self.wgrad_group and mode == 0 and cs >= 64 and not self.use_wgrad_partials and self._wg_pending is not None):
            return False
        if prec == hip.PREC_H3P:                      # plane pairs: whole 64 x 64 channel tiles
            return self.wgrad_group_h3p and cf % 64 == 0 and cs % 64 == 0
        return prec in hip.PREC_16BIT or prec == hip.PREC_H3F_BWD

    def _group_plan(self, B, shapes):
        """Planned layer array, device copy of the 8 item queues, their lengths -- cached per geometry."""
        key = (self.prec_bwd, B, self.wgrad_group_items, shapes)
        plan = self._wg_plans.get(key)
        if plan is None:
            lib = hip.load_library()
            arr = (hip.WgradGroupLayer * len(shapes))()
            for d, (cf, cs, h, w, f_ld, s_ld) in zip(arr, shapes):
                d.CF, d.CS, d.Hf, d.Wf, d.f_ld, d.s_ld = cf, cs, h, w, f_ld, s_ld
            counts = (C.c_int * 8)()
            cap = lib.crimac_wgrad_group_plan(self.prec_bwd, arr, len(shapes), B, self.wgrad_group_items, None, 0, counts)
            if cap < 0:
                raise hip.HipLibraryError("crimac_wgrad_group_plan failed: " + lib.crimac_last_error().decode(errors="replace"))
            items = torch.zeros(8 * max(cap, 1) * 2, dtype=torch.int32)
            rc = lib.crimac_wgrad_group_plan(self.prec_bwd, arr, len(shapes), B, self.wgrad_group_items,
                                             C.c_void_p(items.data_ptr()), max(cap, 1), counts)
            if rc < 0:
                raise hip.HipLibraryError("crimac_wgrad_group_plan failed: " + lib.crimac_last_error().decode(errors="replace"))
            plan = (arr, items.to(self.device), counts, max(cap, 1))
            self._wg_plans[key] = plan
        return plan

    def _flush_wgrad_group(self):
        pend = self._wg_pending
        if not pend:
            return
        self._wg_pending = []
        B = pend[0]["B"]
        shapes = tuple((q["cf"], q["cs"], q["h"], q["w"], q["f_ld"], q["s_ld"]) for q in pend)
        planned, items_dev, counts, cap = self._group_plan(B, shapes)
        arr = (hip.WgradGroupLayer * len(pend))()
        for d, pl, q in zip(arr, planned, pend):
            C.memmove(C.byref(d), C.byref(pl), C.sizeof(hip.WgradGroupLayer))
            d.f, d.s, d.dw = q["f"].value, q["s"].value, q["dw"].data_ptr()
        if self._wg_launches >= self._wg_counters.numel() // 8:
            raise RuntimeError("more grouped weight-gradient launches in one backward pass than queue counters")
        ctr = ptr(self._wg_counters, 8 * self._wg_launches)
        self._wg_launches += 1
        flops = sum(q["flops"] for q in pend)
        args = (self.prec_bwd, C.byref(arr), len(pend), B, ptr(items_dev), cap, C.byref(counts), ctr)
        self._on_side(lambda: call("crimac_wgrad_group", *args, flops=flops, mfmas=hip.MFMAS_PER_PRODUCT[args[0]]))

    def _on_side(self, fn):
        """Run ``fn`` (kernel launches) on the weight-gradient side stream, ordered after everything queued on the
        caller's stream so far; on the caller's stream when the side stream is off (or under gloo rehearsals)."""
        if self.wgrad_side_streams <= 0 or self._gloo_ranks():
            fn()
            return
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(self.wgrad_side_streams)]
            self._side_events = [torch.cuda.Event() for _ in range(32)]
            self._side_i = 0
        ev = self._side_events[self._side_i % len(self._side_events)]
        side = self._side[self._side_i % len(self._side)]
        self._side_i += 1
        ev.record()                                   # everything the launch reads has been queued on this stream
        with torch.cuda.stream(side):
            side.wait_event(ev)
            fn()

    # (ablation runs only, tools/runs/r5_14.sh: 'single' skips the launches that are not part of a grouped launch -- first
    # layer and transposed convolutions --, 'all' skips every weight gradient; gradients are then WRONG, the step time shows
    # what those launches cost the step beyond what the side stream hides)
    _skip_wgrad = os.environ.get("CRIMAC_EXP_SKIP_WGRAD", "")

    def _wgrad(self, prec, mode, f, f_ld, cf, s_, s_ld, cs, B, h, w, key, flops=None):
        dwt, sp, stride = self._dw(key)
        if self._skip_wgrad == "all" or (self._skip_wgrad == "single" and not self._groupable(prec, mode, cf, cs)):
            return
        if self._groupable(prec, mode, cf, cs):
            self._wg_pending.append(dict(f=f, f_ld=f_ld, cf=cf, s=s_, s_ld=s_ld, cs=cs, B=B, h=h, w=w, dw=dwt,
                                         flops=flops or 0.0))
            if len(self._wg_pending) >= min(self.wgrad_group_layers, hip.WGRAD_GROUP_MAX_LAYERS):
                self._flush_wgrad_group()
            return
        if self.use_wgrad_partials:
            name, args = "crimac_wgrad_partials", (prec, mode, f, f_ld, cf, s_, s_ld, cs, B, h, w, ptr(dwt), stride,
                                                   self.wgrad_target_blocks)
        else:
            name, args = "crimac_wgrad", (prec, mode, f, f_ld, cf, s_, s_ld, cs, B, h, w, ptr(dwt),
                                          self.wgrad_target_blocks)
        self._on_side(lambda: call(name, *args, flops=flops, mfmas=hip.MFMAS_PER_PRODUCT[prec]))

    def _join_wgrad(self):
        if self._side is not None:
            for side in self._side:
                torch.cuda.current_stream().wait_stream(side)

    def _unpack_group(self, gi, on_ready=None, rng=None):
        """Packed weight gradients of backward group gi -> torch-layout gradients in the flat buffer; then
        ``on_ready(*rng)`` (the gradient exchange of that range).  With ONE side stream both are queued behind
        the weight gradients on that stream -- the caller's stream is not held up (it joins at the end of the
        backward pass); a collective launched there orders itself after the side stream."""
        self._flush_wgrad_group()                     # the group's conv3x3 weight gradients: one persistent launch
        arr, bounds = self._layer_table()
        first, n = bounds[gi]
        side = self._side[0] if (self._side is not None and len(self._side) == 1 and self.unpack_on_side) else None
        if side is not None:
            # bias / BatchNorm gradients of the range are written on the caller's stream: order them first
            ev = self._side_events[self._side_i % len(self._side_events)]
            self._side_i += 1
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if n:
                    call("crimac_unpack_wgrad_layers", C.byref(arr, first * C.sizeof(hip.LayerDesc)), n)
                if on_ready is not None and self.exchange_on_side:
                    on_ready(*rng)
                done = torch.cuda.Event()
                done.record()
                self._unpacked[gi] = done             # (gradient range gi is final on the side stream from here)
            if on_ready is not None and not self.exchange_on_side:
                torch.cuda.current_stream().wait_stream(side)       # the collective is queued on the caller's stream
                on_ready(*rng)
            return
        self._join_wgrad()
        if n:
            call("crimac_unpack_wgrad_layers", C.byref(arr, first * C.sizeof(hip.LayerDesc)), n)
        if on_ready is not None:
            on_ready(*rng)

    def _pack_ups(self):
        for u in self.ups:
            pk = self.pk[u.key]
            call("crimac_pack_upconv2x2", ptr(self.P[u.key + ".weight"]), u.cin, u.cout, self.planes_arg,
                 ptr(pk["fwd_hi"]), ptr(pk["fwd_lo"]), ptr(pk["dg_hi"]), ptr(pk["dg_lo"]))

    def _pack_eval(self):
        """Fold eval-mode BatchNorm into the conv: W*s, (b-rm)*s+beta, s=g/sqrt(rv+eps)."""
        if not self._eval_pack_dirty:
            return
        with torch.no_grad():
            for b in self.blocks:
                g, be = self.P[b.bn_key + ".weight"], self.P[b.bn_key + ".bias"]
                rm, rv = self.Bf[b.bn_key + ".running_mean"], self.Bf[b.bn_key + ".running_var"]
                s = g * torch.rsqrt(rv + BN_EPS)
                pk = self.pk_eval[b.conv_key]
                pk["bias"].copy_((self.P[b.conv_key + ".bias"] - rm) * s + be)
                call("crimac_pack_conv3x3", ptr(self.P[b.conv_key + ".weight"]), b.cout, b.cin,
                     b.cin_pad, ptr(s.contiguous()), self.planes_arg | (hip.PLANES_FWD_FRAG if pk.get("fwd_frag") else 0),
                     ptr(pk["fwd_hi"]), ptr(pk["fwd_lo"]), None, None)
                # s must stay alive until the kernel ran: same stream, freed memory is stream-ordered
        # up-conv planes are shared with the train pack (no BN behind them)
        if self._train_pack_dirty:
            self._pack_ups()
        self._eval_pack_dirty = False

    # ------------------------------------------------------------------------------------------
    # buffers
    # ------------------------------------------------------------------------------------------
    _buf_prefix = ""             # (eval forward on two streams: each half batch has its own activation buffers)

    def _buf(self, key, shape, dtype=None):
        dtype = dtype or self.act_dtype
        key = self._buf_prefix + key
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def _wfrag_geometry(self, B, H, W):
        """Fragment-major planes are read by the channel-split kernel only, which addresses its input with 32-bit byte
        offsets: a batch whose largest operand tensor (the level-0 concat buffer, 2 * start_filts channels per pixel) reaches
        2 GB falls back to the 8-wave kernel, which reads row-major planes.  Flip every plane's layout flag for such a
        geometry (and back), and have the planes re-packed."""
        es = 4 if self.is_hp else 2
        small = (B * H * W - 1) * (2 * self.sf) * es + 2 * self.sf * es < (1 << 31)
        if small == getattr(self, "_wfrag_on", True):
            return
        self._wfrag_on = small
        for pks in (self.pk_main, self.pk_eval, self.pk16):
            for key, pk in pks.items():
                ok = pk.get("frag_ok") or (self.pk_main.get(key, {}).get("frag_ok", (False, False)) if pks is self.pk_eval else None)
                if ok is None:
                    continue
                if "fwd_frag" in pk:
                    pk["fwd_frag"] = bool(small and ok[0])
                if "dg_frag" in pk:
                    pk["dg_frag"] = bool(small and ok[1])
        self._ltab = self._ltab16 = None
        self._packed_groups = set()
        self._train_pack_dirty = self._eval_pack_dirty = True

    def _geom(self, B, H, W):
        self._wfrag_geometry(B, H, W)
        D = self.depth
        if H % (2 ** (D - 1)) or W % (2 ** (D - 1)):
            raise ValueError(f"H, W must be divisible by {2 ** (D - 1)} (got {H}x{W})")
        return [(H >> i, W >> i, B * (H >> i) * (W >> i)) for i in range(D)]

    # ------------------------------------------------------------------------------------------
    # kernels wrappers
    # ------------------------------------------------------------------------------------------
    conv_impl = os.environ.get("CRIMAC_CONV_IMPL", "halo")     # 'halo' (conv3x3.hip) | 'gather' (igemm.hip)
    fuse_bn_bwd = os.environ.get("CRIMAC_FUSE_BNB", "1") != "0"   # BN-backward sums inside the dgrad conv
    # encoder levels: d(block output) = d(skip) + unpool(d(pooled)) is rebuilt by the BatchNorm-backward apply pass instead
    # of being stored by crimac_unpool_add and read back (crimac_unpool_bn_bwd_apply_replicas).  Default (set in __init__):
    # on where the gradient is stored in 4 bytes (h3p / h3f / fp32 modes: -0.10 to -0.17 ms of an 18 ms h3f step); off for
    # 16-bit storage, where the sums-only pass is bound by its arithmetic, not by the bytes it no longer writes (bf16:
    # 11.39-11.44 -> 11.42-11.46 ms, same box).  CRIMAC_FUSE_UNPOOL_APPLY=0/1 forces it.
    fuse_unpool_apply = None
    fuse_eval_pool = os.environ.get("CRIMAC_FUSE_EVAL_POOL", "1") != "0"   # eval: max-pool in the conv epilogue
    fuse_up_bnb = os.environ.get("CRIMAC_FUSE_UPBNB", "1") != "0"  # ... and inside the transposed-conv dgrad
    # BatchNorm+ReLU of the last decoder block applied inside the 1x1 head (needs fuse_bn_bwd: the head's backward
    # rebuilds its input from the y it reads for the fused sums)
    fuse_head_bn = fuse_bn_bwd and os.environ.get("CRIMAC_FUSE_HEADBN", "1") != "0"
    wfrag = os.environ.get("CRIMAC_WFRAG", "1") != "0"      # fragment-major weight planes for the channel-split kernel (16-bit modes)
    # rows form of the channel-split kernel (CRIMAC_EPI_WROWS): "0" off, "64" the 64-output-channel launches that the tall form
    # runs otherwise, "all" every legal layer (A/B runs: slower than the 128-channel form wherever that one applies)
    conv_rows = os.environ.get("CRIMAC_CONV_ROWS", "0")

    def _rows_choice(self, b, has_dg):
        """(forward, input gradient) of block b on the rows form.  Not where the persistent 64 -> 64 kernel runs (it reads
        row-major planes): the 64 -> 64 layers and the two 64-channel d(concat) halves of the last decoder block."""
        if self.conv_rows not in ("64", "all") or self.conv_impl != "halo" or (b.cin == 64 and b.cout == 64):
            return False, False
        fwd = b.cout % 64 == 0 and b.cin_pad % 64 == 0
        dg = has_dg and b.cin % 64 == 0 and b.cout % 64 == 0 and not (b.cin == 2 * b.cout and b.cin // 2 == 64)
        if self.conv_rows == "64":
            fwd, dg = fwd and b.cout == 64, dg and b.cin == 64
        return bool(fwd), bool(dg)

    def _conv3x3(self, x: Act, pk, bias, out: Act, B, H, W, cin, cout, relu, dgrad=False, cin_real=None,
                 stats=None, bnb=None, cols=None, out_planes=False, stat_reps=None):
        """3x3 conv (forward planes, or dgrad planes).  ``stats=(sum, sumsq)``: fused statistics of
        the stored output (stat_mode 1).  ``bnb=(block, y)``: the output is the ``da`` of that
        BatchNorm block -> its backward sums are fused in (stat_mode 2, into the block's replica
        accumulators).  ``out_planes`` (h3p): the output feeds another contraction -> stored as fp16 plane pairs
        instead of fp32.  Returns True if the requested fusion ran inside the conv kernel."""
        flops = 2.0 * 9 * (cin_real or cin) * cout * B * H * W
        relu = (hip.EPI_RELU if relu else 0) | (hip.EPI_OUT_PLANES if (out_planes and self.is_hp) else 0)
        if self.is_hp and cin_real is not None and cin_real <= 4 and cin == CIN_PAD and not dgrad:
            relu |= hip.EPI_CIN4          # (the network input: channels 4 .. 15 of the padded pixel are zero)
        w_hi, w_lo = ptr(pk["dg_hi" if dgrad else "fwd_hi"]), ptr(pk["dg_lo" if dgrad else "fwd_lo"])
        if pk.get("dg_frag" if dgrad else "fwd_frag"):
            relu |= hip.EPI_WFRAG         # (the plane was packed fragment-major: _alloc_static)
            if pk.get("dg_rows" if dgrad else "fwd_rows"):
                relu |= hip.EPI_WROWS
        prec = self.prec_bwd if dgrad else self.prec
        if self.conv_impl == "halo":
            mode, s0, s1, by, by_ld, bvec = 0, None, None, None, 0, None
            reps = stat_reps or self._nrep(cout)      # (BatchNorm sums of the cout-channel layer this output belongs to)
            if stats is not None:
                mode, s0, s1 = 1, ptr(stats[0]), ptr(stats[1])
            elif bnb is not None and self.fuse_bn_bwd:
                blk, y = bnb
                mode, s0, s1 = 2, ptr(self._stat(blk, 0)), ptr(self._stat(blk, 1))
                by, by_ld, bvec = y.p, y.ld, ptr(self._bnf(blk, 0))
            if cols is not None:                      # a range of the output channels (crimac_conv3x3_cols)
                call("crimac_conv3x3_cols", prec, x.p, x.ld, B, H, W, cin, cout, w_hi, w_lo, ptr(bias),
                     out.p, out.ld, relu, mode, s0, s1, reps, by, by_ld, bvec, self.cmax,
                     cols[0], cols[1], flops=flops * cols[1] / cout, mfmas=hip.MFMAS_PER_PRODUCT[prec])
                return mode != 0
            call("crimac_conv3x3", prec, x.p, x.ld, B, H, W, cin, cout, w_hi, w_lo, ptr(bias),
                 out.p, out.ld, relu, mode, s0, s1, reps, by, by_ld, bvec, self.cmax,
                 flops=flops, mfmas=hip.MFMAS_PER_PRODUCT[prec])
            return mode != 0
        if self.is_hp:
            raise NotImplementedError("precision 'h3p' runs on the halo convolution kernels only (CRIMAC_CONV_IMPL=halo)")
        call("crimac_igemm_conv", prec, x.p, x.ld, B, H, W, H, W, cin, cout, 9, 3, 1, 1, w_hi, w_lo,
             ptr(bias), cout, out.p, out.ld, relu, 0, 0, flops=flops, mfmas=hip.MFMAS_PER_PRODUCT[prec])
        if stats:
            call("crimac_colstats", self.prec, out.p, out.ld, B * H * W, cout, ptr(stats[0]), ptr(stats[1]))
            return True
        return False

    def _bnb_args(self, blk, y):
        """The seven trailing arguments of a producer kernel that also takes the BatchNorm-backward sums of the
        block its output feeds (y: that block's saved conv output; None: no fusion)."""
        if y is None:
            return (None, 0, None, 0, None, None, 1)
        return (y.p, y.ld, ptr(self._bnf(blk, 0)), self.cmax, ptr(self._stat(blk, 0)), ptr(self._stat(blk, 1)),
                self._nrep(blk.cout))

    def _upconv_fwd(self, x: Act, u, out: Act, B, H, W):
        pk = self.pk[u.key]
        call("crimac_igemm_conv", self.prec, x.p, x.ld, B, H, W, H, W, u.cin, 4 * u.cout, 1, 1, 0, 1,
             ptr(pk["fwd_hi"]), ptr(pk["fwd_lo"]), ptr(self.P[u.key + ".bias"]), u.cout, out.p, out.ld,
             hip.EPI_OUT_PLANES if self.is_hp else 0, 1, u.cout, flops=2.0 * 4 * u.cin * u.cout * B * H * W,
             mfmas=hip.MFMAS_PER_PRODUCT[self.prec])

    def _upconv_dgrad(self, dy: Act, u, out: Act, B, H, W, next_bn=None):
        """dy on the fine grid [B,2H,2W,cout] -> dx on the coarse grid [B,H,W,cin].

        next_bn=(block, y): dx is the ``da`` of that BatchNorm block -> take its backward sums in the epilogue
        (bf16 kernel shapes only).  Returns whether they were taken."""
        pk = self._pk_bwd(u.key)
        hp_in = self.is_hp and not self.bwd16          # (dy is a plane-pair tensor: 8 bytes per element of halves)
        prec = self.prec_bwd if self.bwd16 else self.prec
        if (next_bn is not None and self.fuse_bn_bwd and self.fuse_up_bnb and self.lds_dma
                and u.cout % (32 if hp_in else 64) == 0 and u.cin % 128 == 0
                and (16 if hp_in else 8) * B * H * W * dy.ld < (1 << 31)):
            blk, y = next_bn
            call("crimac_upconv2x2_dgrad_bnb_prec", prec, dy.p, dy.ld, B, H, W, u.cout, u.cin, ptr(pk["dg_hi"]),
                 out.p, out.ld,
                 *self._bnb_args(blk, y), flops=2.0 * 4 * u.cin * u.cout * B * H * W, mfmas=hip.MFMAS_PER_PRODUCT[prec])
            return True
        if self.bwd16:
            raise NotImplementedError("h3f: transposed-convolution shape outside the fused input-gradient kernel "
                                      f"(Cout={u.cout} % 64, Cin={u.cin} % 128); precision 'h3p' covers it")
        self._upconv_dgrad_plain(dy, u, out, B, H, W)
        return False

    def _upconv_dgrad_plain(self, dy: Act, u, out: Act, B, H, W):
        pk = self.pk[u.key]
        call("crimac_igemm_conv", self.prec_bwd, dy.p, dy.ld, B, 2 * H, 2 * W, H, W, u.cout, u.cin, 4, 2, 0,
             2, ptr(pk["dg_hi"]), ptr(pk["dg_lo"]), None, 0, out.p, out.ld, 0, 0, 0,
             flops=2.0 * 4 * u.cin * u.cout * B * H * W, mfmas=hip.MFMAS_PER_PRODUCT[self.prec_bwd])

    # SyncBN (optional, N > 1): BatchNorm statistics over the GLOBAL batch.  Per BN layer one all-reduce of
    # (sum y, sum y^2) [2 x cmax fp64] in the forward pass and one of (sum dz, sum dz*xhat) in the backward
    # pass, on a process group of their own so they do not queue behind the gradient buckets.  With it an
    # N-rank step equals a single-rank step on the concatenated batch (loss = mean of the per-rank losses).
    sync_bn = False
    _bn_group = None

    def _sync_world(self):
        import torch.distributed as dist
        if not (self.sync_bn and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return 1
        if UNetEngine._bn_group is None:
            UNetEngine._bn_group = dist.new_group()
        return dist.get_world_size()

    def _stat_pair(self, b):
        """The two adjacent [cmax] fp64 slots (sum_dz | sum_dz_xhat) of BN layer b."""
        o = 2 + b.idx * (2 * STAT_REPLICAS + 2) * self.cmax + 2 * STAT_REPLICAS * self.cmax
        return self.stat[o:o + 2 * self.cmax]

    def _bn_train(self, b, y: Act, M):
        """Batch statistics were accumulated by the conv epilogue; finish them."""
        world = self._sync_world()
        if world > 1:
            import torch.distributed as dist
            tmp = self._buf("syncbn.fwd", (2, self.cmax), torch.float64)   # (the backward slots must stay zero)
            call("crimac_sum_replicas", ptr(self._stat(b, 0)), self._nrep(b.cout), b.cout, b.cout,
                 ptr(tmp[0]), None, ptr(self._stat(b, 1)), ptr(tmp[1]))
            dist.all_reduce(tmp, group=UNetEngine._bn_group)
            call("crimac_bn_finalize", ptr(tmp[0]), ptr(tmp[1]), 1, M * world, b.cout,
                 ptr(self.P[b.bn_key + ".weight"]), ptr(self.P[b.bn_key + ".bias"]), BN_EPS, BN_MOMENTUM,
                 ptr(self.Bf[b.bn_key + ".running_mean"]), ptr(self.Bf[b.bn_key + ".running_var"]),
                 ptr(self.Bf[b.bn_key + ".num_batches_tracked"]), ptr(self._bnf(b, 0)),
                 ptr(self._bnf(b, 1)), ptr(self._bnf(b, 2)), ptr(self._bnf(b, 3)))
            return
        call("crimac_bn_finalize", ptr(self._stat(b, 0)), ptr(self._stat(b, 1)), self._nrep(b.cout), M, b.cout,
             ptr(self.P[b.bn_key + ".weight"]), ptr(self.P[b.bn_key + ".bias"]), BN_EPS, BN_MOMENTUM,
             ptr(self.Bf[b.bn_key + ".running_mean"]), ptr(self.Bf[b.bn_key + ".running_var"]),
             ptr(self.Bf[b.bn_key + ".num_batches_tracked"]), ptr(self._bnf(b, 0)),
             ptr(self._bnf(b, 1)), ptr(self._bnf(b, 2)), ptr(self._bnf(b, 3)))

    def _bn_act(self, b, y: Act, out: Act, pool: Act, B, H, W, M):
        """Train-mode BatchNorm + ReLU (+ max-pool) of the conv output y whose statistics the conv epilogue took."""
        if not self.fold_bn_finalize or self._sync_world() > 1:
            self._bn_train(b, y, M)
            self._act(b, y, out, pool, B, H, W)
            return
        call("crimac_bn_train_act_pool", self.prec, y.p, y.ld, ptr(self._stat(b, 0)), ptr(self._stat(b, 1)),
             self._nrep(b.cout), M, ptr(self.P[b.bn_key + ".weight"]), ptr(self.P[b.bn_key + ".bias"]), BN_EPS,
             BN_MOMENTUM, ptr(self.Bf[b.bn_key + ".running_mean"]), ptr(self.Bf[b.bn_key + ".running_var"]),
             ptr(self.Bf[b.bn_key + ".num_batches_tracked"]), ptr(self._bnf(b, 0)), self.cmax, 1,
             out.p if out is not None else None, out.ld if out is not None else 0,
             pool.p if pool is not None else None, pool.ld if pool is not None else 0, B, H, W, b.cout)

    def _act(self, b, y: Act, out: Act, pool: Act, B, H, W, train=True):
        call("crimac_bn_act_pool", self.prec, y.p, y.ld,
             ptr(self._bnf(b, 2)) if train else None, ptr(self._bnf(b, 3)) if train else None,
             1 if train else 0, out.p if out is not None else None, out.ld if out is not None else 0,
             pool.p if pool is not None else None, pool.ld if pool is not None else 0, B, H, W, b.cout)

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _input(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input [B,{self.in_channels},H,W], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise hip.HipLibraryError("input is not on a GPU: the HIP path has no CPU fallback")
        x = x.contiguous().float()
        B, _, H, W = x.shape
        xin = self._buf("x_nhwc", (B * H * W, CIN_PAD))
        call("crimac_nchw_to_nhwc", self.prec, ptr(x), ptr(xin), B, self.in_channels, H, W, CIN_PAD)
        return xin, B, H, W

    @_on_device
    def forward(self, x, training, softmax=False, meta=None):
        """Logits [B,n_classes,H,W] fp32 (NCHW).  Train mode keeps what backward needs.
        ``meta`` [B,Cm,H,W]: the metadata planes of UNet_LateMetInject.forward(x, meta_tensor) (unet.py:372)."""
        self.bind()
        if training and self.leader is not None:
            raise RuntimeError("this engine only runs eval-mode forwards (it shares the parameters of the model's training "
                               "engine); train-mode passes go through model.engine")
        if self.lmi:
            meta = self._meta(meta, x)
        elif meta is not None:
            raise ValueError("this model takes no metadata tensor (late_meta_inject=False)")
        if (not training and self.eval_two_streams and x.dim() == 4 and x.shape[0] >= 16
                and x.is_cuda and not self._gloo_ranks() and not self.lmi):
            return self._forward_eval_two_streams(x, softmax)
        xin, B, H, W = self._input(x)
        return self.forward_nhwc(xin, B, H, W, training, softmax, meta=meta)

    def _meta(self, meta, x):
        if meta is None:
            raise ValueError("UNet_LateMetInject.forward needs the metadata tensor")
        if not meta.is_cuda:
            raise hip.HipLibraryError("metadata tensor is not on a GPU: the HIP path has no CPU fallback")
        if meta.dim() != 4 or meta.shape[1] != self.meta_channels or meta.shape[0] != x.shape[0] \
                or tuple(meta.shape[2:]) != tuple(x.shape[2:]):
            raise ValueError(f"expected metadata [B,{self.meta_channels},H,W] matching the data, got {tuple(meta.shape)}")
        return meta.contiguous().float()

    def _lmi_weights(self):
        """conv_final.weight [nc, 65, 1, 1] -> (first 64 columns packed [nc, 64], metadata column [nc])"""
        w = self.P["conv_final.weight"].detach().view(self.n_classes, self.sf + 1)
        w64 = self._buf("lmi.w64", (self.n_classes, self.sf), torch.float32)
        wm = self._buf("lmi.wm", (self.n_classes,), torch.float32)
        w64.copy_(w[:, :self.sf])
        wm.copy_(w[:, self.sf])
        return w64, wm

    def _mlp(self):
        pre = "post_processing_weights.main."
        return [ptr(self.P[pre + k]) for k in ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias")]

    # Inference: the two halves of a batch run on two streams (own activation buffers each).  The kernels of one
    # half fill the ragged last rounds and the launch gaps of the other -- the same effect the side stream has in
    # the backward pass.
    eval_two_streams = os.environ.get("CRIMAC_EVAL_STREAMS", "1") != "0"

    def _forward_eval_two_streams(self, x, softmax):
        # (a stream of its own: the backward pass's side-stream logic keys on self._side, which must stay None when
        # CRIMAC_WGRAD_STREAM=0 asks for a serialized backward pass or under gloo rehearsals)
        if self._eval_side is None:
            self._eval_side = torch.cuda.Stream(device=self.device)
            self._eval_events = [torch.cuda.Event() for _ in range(8)]
            self._eval_i = 0
        side = self._eval_side
        B, _, H, W = x.shape
        h = B // 2
        self._check_versions()
        self._pack_eval()                              # (on the caller's stream, before either half starts)
        logits = torch.empty((B, self.n_classes, H, W), dtype=torch.float32, device=self.device)
        ev = self._eval_events[self._eval_i % len(self._eval_events)]
        self._eval_i += 1
        ev.record()
        try:
            self._buf_prefix = "h0."
            xin, _, _, _ = self._input(x[:h])
            self.forward_nhwc(xin, h, H, W, False, softmax, out=logits[:h])
            self._buf_prefix = "h1."
            with torch.cuda.stream(side):
                side.wait_event(ev)
                xin, _, _, _ = self._input(x[h:])
                self.forward_nhwc(xin, B - h, H, W, False, softmax, out=logits[h:])
        finally:
            self._buf_prefix = ""
        torch.cuda.current_stream().wait_stream(side)
        return logits

    @_on_device
    def forward_nhwc_eval_split(self, xin, B, H, W, softmax=False):
        """Eval forward from an NHWC matrix (tiled inference), the batch split over two streams like
        ``_forward_eval_two_streams`` (uneven halves allowed)."""
        self.bind()
        if not (self.eval_two_streams and B >= 16 and not self._gloo_ranks() and not self.lmi):
            return self.forward_nhwc(xin, B, H, W, False, softmax)
        if self._eval_side is None:
            self._eval_side = torch.cuda.Stream(device=self.device)
            self._eval_events = [torch.cuda.Event() for _ in range(8)]
            self._eval_i = 0
        side = self._eval_side
        h = B // 2
        self._check_versions()
        self._pack_eval()
        logits = torch.empty((B, self.n_classes, H, W), dtype=torch.float32, device=self.device)
        ev = self._eval_events[self._eval_i % len(self._eval_events)]
        self._eval_i += 1
        ev.record()
        try:
            self._buf_prefix = "h0."
            self.forward_nhwc(xin[:h * H * W], h, H, W, False, softmax, out=logits[:h])
            self._buf_prefix = "h1."
            with torch.cuda.stream(side):
                side.wait_event(ev)
                self.forward_nhwc(xin[h * H * W:], B - h, H, W, False, softmax, out=logits[h:])
        finally:
            self._buf_prefix = ""
        torch.cuda.current_stream().wait_stream(side)
        return logits

    @_on_device
    def forward_nhwc(self, xin, B, H, W, training, softmax=False, out=None, meta=None):
        """Same, from an NHWC activation matrix [B*H*W, 16] already in the engine's storage type
        (what ``crimac_gather_patches`` writes for the tiled-inference path)."""
        self.bind()
        self._check_versions()
        if tuple(xin.shape) != (B * H * W, CIN_PAD) or xin.dtype != self.act_dtype or not xin.is_cuda:
            raise ValueError(f"expected a [{B * H * W},{CIN_PAD}] {self.act_dtype} GPU tensor")
        geo = self._geom(B, H, W)
        D = self.depth
        if training:
            if self.leader is not None:
                raise RuntimeError("follower engine: eval-mode forwards only (train through model.engine)")
            self._pack_train()
            self.stat.zero_()
            # the BatchNorm running statistics change with every train-mode forward: eval-mode operands (BN folded into
            # the convolution) of this engine and of its followers are stale from here on
            self._eval_pack_dirty = True
            for f in self._followers:
                f._eval_pack_dirty = True
        else:
            self._pack_eval()
        cur = Act(xin, CIN_PAD)
        saved = {"B": B, "H": H, "W": W, "x0": cur}
        head_bn = None
        for i in range(D):
            h, w, M = geo[i]
            c = self.sf * 2 ** i
            b1, b2 = self.enc[i]
            a1 = Act(self._buf(f"e{i}.a1", (M, c)), c)
            if i < D - 1:
                cat = self._buf(f"cat{i}", (M, 2 * c))
                a2 = Act(cat, c, off=c, ld=2 * c)
                pool = Act(self._buf(f"e{i}.pool", (geo[i + 1][2], c)), c)
            else:
                a2 = Act(self._buf(f"e{i}.a2", (M, c)), c)
                pool = None
            if training:
                y1 = Act(self._buf(f"e{i}.y1", (M, c)), c)
                y2 = Act(self._buf(f"e{i}.y2", (M, c)), c)
                self._conv3x3(cur, self.pk[b1.conv_key], self.P[b1.conv_key + ".bias"], y1, B, h, w,
                              b1.cin_pad, c, relu=False, cin_real=b1.cin,
                              stats=(self._stat(b1, 0), self._stat(b1, 1)))
                self._bn_act(b1, y1, a1, None, B, h, w, M)
                self._conv3x3(a1, self.pk[b2.conv_key], self.P[b2.conv_key + ".bias"], y2, B, h, w, c, c,
                              relu=False, stats=(self._stat(b2, 0), self._stat(b2, 1)))
                self._bn_act(b2, y2, a2, pool, B, h, w, M)
                saved[f"e{i}"] = (cur, y1, a1, y2, a2)
            else:
                pe1, pe2 = self.pk_eval[b1.conv_key], self.pk_eval[b2.conv_key]
                self._conv3x3(cur, pe1, pe1["bias"], a1, B, h, w, b1.cin_pad, c, relu=True, cin_real=b1.cin,
                              out_planes=True)
                if pool is not None and self.fuse_eval_pool and self.conv_impl == "halo":
                    # the max-pool comes out of the conv epilogue (the tile is still in LDS)
                    call("crimac_conv3x3_pool", self.prec, a1.p, a1.ld, B, h, w, c, c, ptr(pe2["fwd_hi"]),
                         ptr(pe2["fwd_lo"]), ptr(pe2["bias"]), a2.p, a2.ld,
                         hip.EPI_RELU | (hip.EPI_OUT_PLANES if self.is_hp else 0)
                         | (hip.EPI_WFRAG if pe2.get("fwd_frag") else 0)
                         | (hip.EPI_WROWS if (pe2.get("fwd_frag") and pe2.get("fwd_rows")) else 0), pool.p, pool.ld,
                         flops=2.0 * 9 * c * c * B * h * w, mfmas=hip.MFMAS_PER_PRODUCT[self.prec])
                else:
                    self._conv3x3(a1, pe2, pe2["bias"], a2, B, h, w, c, c, relu=True, out_planes=True)
                    if pool is not None:
                        self._act(b2, a2, None, pool, B, h, w, train=False)
            cur = pool if pool is not None else a2
        for j in range(D - 1):
            L = D - 2 - j
            h, w, M = geo[L]
            c = self.sf * 2 ** L
            u = self.ups[j]
            b1, b2 = self.dec[j]
            cat = self._buf(f"cat{L}", (M, 2 * c))
            up = Act(cat, c, off=0, ld=2 * c)
            catA = Act(cat, 2 * c)
            hp, wp, _ = geo[L + 1]
            self._upconv_fwd(cur, u, up, B, hp, wp)
            a1 = Act(self._buf(f"d{j}.a1", (M, c)), c)
            a2 = Act(self._buf(f"d{j}.a2", (M, c)), c)
            if training:
                y1 = Act(self._buf(f"d{j}.y1", (M, c)), c)
                y2 = Act(self._buf(f"d{j}.y2", (M, c)), c)
                self._conv3x3(catA, self.pk[b1.conv_key], self.P[b1.conv_key + ".bias"], y1, B, h, w,
                              2 * c, c, relu=False, stats=(self._stat(b1, 0), self._stat(b1, 1)))
                self._bn_act(b1, y1, a1, None, B, h, w, M)
                self._conv3x3(a1, self.pk[b2.conv_key], self.P[b2.conv_key + ".bias"], y2, B, h, w, c, c,
                              relu=False, stats=(self._stat(b2, 0), self._stat(b2, 1)))
                if j == D - 2 and self.fuse_head_bn and self.fuse_bn_bwd:
                    # the last block's activation only feeds the 1x1 head: formed on the fly there (forward and
                    # backward) from y2 -- one read + one write of the largest activation saved, twice
                    self._bn_train(b2, y2, M)
                    head_bn = b2
                    a2 = y2
                else:
                    self._bn_act(b2, y2, a2, None, B, h, w, M)
                saved[f"d{j}"] = (cur, catA, y1, a1, y2, a2)
            else:
                pe1, pe2 = self.pk_eval[b1.conv_key], self.pk_eval[b2.conv_key]
                self._conv3x3(catA, pe1, pe1["bias"], a1, B, h, w, 2 * c, c, relu=True, out_planes=True)
                self._conv3x3(a1, pe2, pe2["bias"], a2, B, h, w, c, c, relu=True, out_planes=True)
            cur = a2
        logits = out if out is not None else torch.empty((B, self.n_classes, H, W), dtype=torch.float32,
                                                         device=self.device)
        if self.lmi:
            if meta is None:
                raise ValueError("UNet_LateMetInject needs the metadata tensor")
            w64, wm = self._lmi_weights()
            call("crimac_head_fwd", self.prec, cur.p, cur.ld, self.sf, ptr(w64), ptr(self.P["conv_final.bias"]),
                 ptr(logits), B, H, W, self.n_classes, 0,
                 ptr(self._bnf(head_bn, 2)) if head_bn is not None else None,
                 ptr(self._bnf(head_bn, 3)) if head_bn is not None else None)
            m = self._buf("lmi.m", (B * H * W,), torch.float32)
            call("crimac_meta_mlp_fwd", ptr(meta), self.meta_channels, B, H, W, *self._mlp(), ptr(m))
            call("crimac_meta_inject_fwd", ptr(m), ptr(wm), ptr(logits), B, H, W, self.n_classes, 1 if softmax else 0)
            saved["meta"] = meta
        else:
            call("crimac_head_fwd", self.prec, cur.p, cur.ld, self.sf, ptr(self.P["conv_final.weight"]),
                 ptr(self.P["conv_final.bias"]), ptr(logits), B, H, W, self.n_classes, 1 if softmax else 0,
                 ptr(self._bnf(head_bn, 2)) if head_bn is not None else None,
                 ptr(self._bnf(head_bn, 3)) if head_bn is not None else None)
        saved["head_in"] = cur if head_bn is None else None
        self.saved = saved if training else None
        if training:
            self.forward_generation += 1
        return logits

    # ------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------
    def _block_bwd(self, tag, b, da: Act, y: Act, x_in: Act, B, h, w, M, dx_out: Act, reduce_done=False,
                   next_bn=None, bias_from_stats=None, unpool_src=None):
        """Backward of conv3x3+BN+ReLU given da (grad of the block output).

        reduce_done: the producer of ``da`` already accumulated this block's BatchNorm-backward sums
        into its replica accumulators (conv3x3 stat_mode 2).  next_bn=(block, y): ``dx_out`` is the
        ``da`` of that block -> fuse ITS sums into the dgrad convolution.  bias_from_stats=(grad, C, level):
        also take the per-channel sums of the first C channels of ``dx_out`` (a transposed-conv bias
        gradient) from the dgrad epilogue.  unpool_src=(dp, ds): ``da`` was NOT stored -- it is ds + unpool(dp), whose
        BatchNorm-backward sums crimac_unpool_add took (sums-only call); the apply kernel rebuilds it.
        Returns whether next_bn's sums were fused."""
        world = self._sync_world()
        folded = reduce_done and self.fold_bn_finalize and world == 1     # the replicas are added up by the apply kernel
        if folded:
            pass
        elif reduce_done:
            call("crimac_sum_replicas", ptr(self._stat(b, 0)), self._nrep(b.cout), b.cout, b.cout,
                 ptr(self._stat(b, 2)), None, ptr(self._stat(b, 1)), ptr(self._stat(b, 3)))
        else:
            call("crimac_bn_bwd_reduce", self.prec, da.p, da.ld, y.p, y.ld, ptr(self._bnf(b, 2)),
                 ptr(self._bnf(b, 3)), ptr(self._bnf(b, 0)), ptr(self._bnf(b, 1)), M, b.cout,
                 ptr(self._stat(b, 2)), ptr(self._stat(b, 3)))
        # (h3f: dy is an MFMA operand of the backward pass only -> plain fp16; da and y stay fp32)
        dy = Act(self._buf(f"{tag}.dy", (M, b.cout), torch.float16 if self.bwd16 else None), b.cout)
        prec_apply = self.prec_bwd if self.bwd16 else self.prec
        dgamma, dbeta = self.G[b.bn_key + ".weight"], self.G[b.bn_key + ".bias"]
        if world > 1:
            # gamma / beta gradients are this rank's LOCAL sums (the gradient exchange adds the ranks up); the
            # input gradient needs the GLOBAL sums and the global pixel count
            import torch.distributed as dist
            dgamma.copy_(self._stat(b, 3))
            dbeta.copy_(self._stat(b, 2))
            dist.all_reduce(self._stat_pair(b), group=UNetEngine._bn_group)
            scr = self._buf("g.syncbn.scratch", (2, self.cmax), torch.float32)
            dgamma, dbeta = scr[0], scr[1]
        # (d(conv bias in front of train-mode BN) = sum dy == 0 exactly: left at the zero fill -- the reference holds
        # ~1e-8 rounding noise there; 2048 x C same-address atomics saved)
        if unpool_src is not None:
            if not folded:
                raise RuntimeError("unpool_src needs the folded BatchNorm-backward path")
            dp_, ds_ = unpool_src
            call("crimac_unpool_bn_bwd_apply_replicas", prec_apply, dp_.p, dp_.ld, ds_.p if ds_ is not None else None,
                 ds_.ld if ds_ is not None else 0, y.p, y.ld, ptr(self._bnf(b, 0)), self.cmax, ptr(self._stat(b, 0)),
                 ptr(self._stat(b, 1)), self._nrep(b.cout), M, dy.p, dy.ld, B, h, w, b.cout, ptr(dgamma), ptr(dbeta))
        elif folded:
            call("crimac_bn_bwd_apply_replicas", prec_apply, da.p, da.ld, y.p, y.ld, ptr(self._bnf(b, 0)), self.cmax,
                 ptr(self._stat(b, 0)), ptr(self._stat(b, 1)), self._nrep(b.cout), M, M, b.cout, dy.p, dy.ld,
                 ptr(dgamma), ptr(dbeta))
        else:
            call("crimac_bn_bwd_apply", prec_apply, da.p, da.ld, y.p, y.ld, ptr(self._bnf(b, 2)),
                 ptr(self._bnf(b, 3)), ptr(self._bnf(b, 0)), ptr(self._bnf(b, 1)), ptr(self._stat(b, 2)),
                 ptr(self._stat(b, 3)), M, M * world, b.cout, dy.p, dy.ld, ptr(dgamma), ptr(dbeta), None)
        self._wgrad(self.prec_bwd, 0, dy.p, dy.ld, b.cout, x_in.p, x_in.ld, b.cin_pad, B, h, w, b.conv_key,
                    flops=2.0 * 9 * b.cin * b.cout * B * h * w)
        fused = False
        if dx_out is not None:
            stats = None
            if bias_from_stats is not None and self.conv_impl == "halo":
                per = 2 * STAT_REPLICAS * 2 * self.cmax                  # this level's slice (zeroed in backward())
                lvl = bias_from_stats[2]
                scr = self.bias_scr[lvl * per:(lvl + 1) * per]
                stats = (scr[:per // 2], scr[per // 2:])
            C_up = bias_from_stats[1] if bias_from_stats is not None else 0
            side = self._side[0] if (self._side is not None and len(self._side) == 1) else None
            side_ok = side is not None and self.split_skip_dgrad and not self._gloo_ranks()
            if self.is_hp and bias_from_stats is not None:
                # plane pairs: the up half of d(concat) feeds two contractions (transposed-conv input and weight
                # gradients) -> plane pairs; the skip half is read by unpool_add -> fp32: always two launches
                if stats is None or b.cin != 2 * C_up or C_up % 64 != 0:
                    raise NotImplementedError("h3p: decoder convolution shape outside the plane-pair kernels")
                dx_up = dx_out
                if self.bwd16:
                    # h3f: the up half is an fp16 operand (2-byte elements) -- a buffer of its own, not a slice of the
                    # 4-byte-addressed d(concat) buffer
                    dx_up = Act(self._buf(f"{tag}.dup16", (M, C_up), torch.float16), C_up)
                    self._dup16[bias_from_stats[2]] = dx_up
                self._conv3x3(dy, self._pk_bwd(b.conv_key), None, dx_up, B, h, w, b.cout, b.cin, relu=False,
                              dgrad=True, stats=stats, cols=(0, C_up), out_planes=True)
                if side_ok:
                    ev = self._side_events[self._side_i % len(self._side_events)]
                    self._side_i += 1
                    ev.record()
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        self._conv3x3(dy, self._pk_bwd(b.conv_key), None, dx_out, B, h, w, b.cout, b.cin, relu=False,
                                      dgrad=True, cols=(C_up, C_up))
                        done = torch.cuda.Event()
                        done.record()
                    self._skip_done[bias_from_stats[2]] = done
                else:
                    self._conv3x3(dy, self._pk_bwd(b.conv_key), None, dx_out, B, h, w, b.cout, b.cin, relu=False,
                                  dgrad=True, cols=(C_up, C_up))
                fused = True
            elif (stats is not None and b.cin == 2 * C_up and b.cout % 64 == 0 and self.is16
                    and ((side_ok and C_up % 128 == 0)
                         or (C_up == 64 and b.cout == 64 and B * ((h + 15) // 16) * ((w + 15) // 16) >= 512))):
                # decoder conv1: dx_out = d(concat [up | skip]).  The up half (and its column sums = the transposed
                # convolution's bias gradient) is needed at once; the skip half only when the encoder level is
                # reached -> side stream, off the critical path.  (The last decoder level -- two 64-channel halves of a
                # 64-channel dy at 256 x 256 -- takes the two launches of the persistent 64 -> 64 kernel with or without a
                # side stream: 2 x ~150 us against 330-344 us for the one-chunk launch of the channel-split kernel, and the
                # serialized roofline pass then times the kernels the overlapped step runs.)
                self._conv3x3(dy, self._pk_bwd(b.conv_key), None, dx_out, B, h, w, b.cout, b.cin, relu=False,
                              dgrad=True, stats=stats, cols=(0, C_up))
                if side_ok:
                    ev = self._side_events[self._side_i % len(self._side_events)]
                    self._side_i += 1
                    ev.record()
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        self._conv3x3(dy, self._pk_bwd(b.conv_key), None, dx_out, B, h, w, b.cout, b.cin, relu=False,
                                      dgrad=True, cols=(C_up, C_up))
                        done = torch.cuda.Event()
                        done.record()
                    self._skip_done[bias_from_stats[2]] = done
                else:
                    self._conv3x3(dy, self._pk_bwd(b.conv_key), None, dx_out, B, h, w, b.cout, b.cin, relu=False,
                                  dgrad=True, cols=(C_up, C_up))
                fused = True
            else:
                fused = self._conv3x3(dy, self._pk_bwd(b.conv_key), None, dx_out, B, h, w, b.cout, b.cin, relu=False,
                                      dgrad=True, stats=stats, bnb=next_bn if stats is None else None)
            if stats is not None:
                grad, C = bias_from_stats[:2]
                call("crimac_sum_replicas", ptr(stats[0]), STAT_REPLICAS, b.cin, C, None, ptr(grad), None, None)
            elif bias_from_stats is not None:
                grad, C = bias_from_stats[:2]
                call("crimac_colsum_f32", self.prec, dx_out.p, dx_out.ld, M, C, ptr(grad))
        return fused and next_bn is not None

    wgrad_target_blocks = 0      # 0 = let the library pick the pixel-range split

    # Backward completes the flat gradient buffer from its end: decoder + head first (group 0), then the
    # encoder blocks deepest first.  The two deepest encoder blocks get a group each (14.2 M and 3.5 M of the
    # 31 M gradients), the shallow rest (1.1 M) goes last -- that tail is all that is exchanged after the
    # backward pass has finished.
    n_enc_groups = 3

    def _enc_group(self, i):
        """Backward group (1..n_enc_groups) of encoder block i."""
        return min(self.depth - i, self.n_enc_groups)

    def grad_ranges(self):
        """Contiguous ranges of the flat gradient buffer in the order the backward pass completes them."""
        if getattr(self, "_granges", None) is not None and self._granges[0] is self.layout:
            return self._granges[1]
        r = self._grad_ranges()
        self._granges = (self.layout, r)
        return r

    def _grad_ranges(self):
        dec = self.ups[0].key.split(".")[0] + "."
        lo_dec = min(o for k, (o, _, _) in self.layout.items() if k.startswith(dec))
        starts = {}                                   # group -> lowest offset
        for i in range(self.depth):
            pre = ".".join(self.enc[i][0].conv_key.split(".")[:2]) + "."
            lo = min(o for k, (o, _, _) in self.layout.items() if k.startswith(pre))
            g = self._enc_group(i)
            starts[g] = min(starts.get(g, lo), lo)
        bounds = sorted(starts.items(), key=lambda kv: -kv[1])      # deepest group (highest offset) first
        if [g for g, _ in bounds] != sorted(starts) or bounds[-1][1] != 0 or not bounds[0][1] < lo_dec:
            raise RuntimeError("unexpected parameter order in the flat buffer")
        for k, (o, _, _) in self.layout.items():
            if (o >= lo_dec) != (k.startswith(dec) or k.startswith("conv_final") or k.startswith("post_processing_weights")):
                raise RuntimeError(f"unexpected parameter order in the flat buffer at {k}")
        ranges, hi = [(lo_dec, self.n_flat)], lo_dec
        for _, lo in bounds:
            ranges.append((lo, hi))
            hi = lo
        return ranges

    @_on_device
    def backward(self, dlogits, on_ready=None, before_join=None):
        """Gradients of every parameter into the flat gradient buffer (overwrites it).

        on_ready(lo, hi): called when flat_g[lo:hi] is final (see grad_ranges) so that the gradient
        exchange of that range can start while the rest of the backward pass runs."""
        ranges = self.grad_ranges() if on_ready is not None else None
        s = self.saved
        if s is None:
            raise RuntimeError("backward() needs a preceding train-mode forward()")
        B, H, W = s["B"], s["H"], s["W"]
        geo = self._geom(B, H, W)
        D = self.depth
        self._skip_done = {}
        self._unpacked = {}
        self._wg_pending, self._wg_launches = [], 0
        self._dup16 = {}
        self._plan_dw(B, H, W)
        self.flat_g.zero_()
        if not self.use_wgrad_partials:
            self.dw_packed.zero_()                    # (the atomic form accumulates; the slab form overwrites)
        # the fused BatchNorm-backward reductions accumulate into the replica slots the forward statistics used:
        # one fill for all layers (and one for the transposed-conv bias sums) instead of one per fused block
        self.stat[2:].zero_()
        self.bias_scr.zero_()
        dlogits = dlogits.contiguous().float()
        head_in = s["head_in"]
        h, w, M = geo[0]
        d_cur = Act(self._buf("g.head", (M, self.sf)), self.sf)
        # d_cur is the `da` of the last decoder block's second BatchNorm: its backward sums are taken here
        head_fused = self.fuse_bn_bwd and D >= 2
        if head_in is None and not head_fused:
            raise RuntimeError("head input was not materialised (fuse_head_bn) but the fused BatchNorm sums are off")
        w_head, g_head = ptr(self.P["conv_final.weight"]), self.G["conv_final.weight"]
        if self.lmi:
            # metadata column + perceptron first (they only need dlogits), then the 64-channel head on the packed columns
            w64, wm = self._lmi_weights()
            g64 = self._buf("lmi.g64", (self.n_classes, self.sf), torch.float32)
            gwm = self._buf("lmi.gwm", (self.n_classes,), torch.float32)
            g64.zero_()
            gwm.zero_()
            pre = "post_processing_weights.main."
            call("crimac_meta_bwd", ptr(dlogits), ptr(s["meta"]), self.meta_channels, B, H, W, self.n_classes, ptr(wm),
                 *self._mlp(), ptr(gwm), *[ptr(self.G[pre + k]) for k in
                                          ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias")])
            w_head, g_head = ptr(w64), g64
        call("crimac_head_bwd", self.prec, ptr(dlogits), head_in.p if head_in is not None else None,
             head_in.ld if head_in is not None else 0, self.sf,
             w_head, d_cur.p, d_cur.ld, ptr(g_head),
             ptr(self.G["conv_final.bias"]), B, H, W, self.n_classes,
             *self._bnb_args(self.dec[D - 2][1], s[f"d{D - 2}"][4] if head_fused else None))
        if self.lmi:
            gw = self.G["conv_final.weight"].view(self.n_classes, self.sf + 1)
            gw[:, :self.sf].copy_(g64)
            gw[:, self.sf].copy_(gwm)
        skip_grad = {}
        cur_done = head_fused            # BatchNorm-backward sums of d_cur's block already taken by its producer
        for j in reversed(range(D - 1)):
            L = D - 2 - j
            h, w, M = geo[L]
            c = self.sf * 2 ** L
            u = self.ups[j]
            b1, b2 = self.dec[j]
            x_prev, catA, y1, a1, y2, a2 = s[f"d{j}"]
            da1 = Act(self._buf(f"g.d{j}.a1", (M, c)), c)
            fused = self._block_bwd(f"g.d{j}.2", b2, d_cur, y2, a1, B, h, w, M, da1, next_bn=(b1, y1),
                                    reduce_done=cur_done)
            dcat = Act(self._buf(f"g.d{j}.cat", (M, 2 * c)), 2 * c)
            # transposed conv bias gradient (unet.py:130) = column sums of dcat[:, :c]: from the dgrad epilogue
            self._block_bwd(f"g.d{j}.1", b1, da1, y1, catA, B, h, w, M, dcat, reduce_done=fused,
                            bias_from_stats=(self.G[u.key + ".bias"], c, j))
            dup = self._dup16.get(j) or dcat.slice(0, c)
            skip_grad[L] = dcat.slice(c, c)
            hp, wp, Mp = geo[L + 1]
            self._wgrad(self.prec_bwd, 1, x_prev.p, x_prev.ld, u.cin, dup.p, dup.ld, u.cout, B, hp, wp, u.key,
                        flops=2.0 * 4 * u.cin * u.cout * B * hp * wp)
            d_prev = Act(self._buf(f"g.d{j}.xprev", (Mp, u.cin)), u.cin)
            # d_prev is the `da` of the next coarser block (decoder j-1, or the bottleneck encoder block)
            nxt = (self.dec[j - 1][1], s[f"d{j - 1}"][4]) if j > 0 else (self.enc[D - 1][1], s[f"e{D - 1}"][3])
            cur_done = self._upconv_dgrad(dup, u, d_prev, B, hp, wp, next_bn=nxt)
            d_cur = d_prev
        self._unpack_group(0, on_ready, ranges[0] if ranges else None)
        d_pool = None
        for i in reversed(range(D)):
            h, w, M = geo[i]
            c = self.sf * 2 ** i
            b1, b2 = self.enc[i]
            x_in, y1, a1, y2, a2 = s[f"e{i}"]
            unpool_src = None
            if i == D - 1:
                da2 = d_cur
            else:
                ds = skip_grad[i]
                ev = self._skip_done.pop(D - 2 - i, None)
                if ev is not None:                      # skip half of d(concat) was produced on the side stream
                    torch.cuda.current_stream().wait_event(ev)
                # da2 = ds + unpool(d_pool) is consumed by this block's BatchNorm backward only: with the fused sums it
                # is never stored -- the first pass takes the sums, the apply pass rebuilds it (-1.5 of 12.5 bytes per
                # element of the two passes in bf16)
                if (self.fuse_unpool_apply and self.fuse_bn_bwd and self.fold_bn_finalize
                        and self._sync_world() == 1):
                    da2, unpool_src = None, (d_pool, ds)
                else:
                    da2 = Act(self._buf(f"g.e{i}.a2", (M, c)), c)
                call("crimac_unpool_add", self.prec, d_pool.p, d_pool.ld, a2.p, a2.ld, ds.p, ds.ld,
                     da2.p if da2 is not None else None, da2.ld if da2 is not None else 0, B, h, w, c,
                     *self._bnb_args(b2, y2 if self.fuse_bn_bwd else None))
            da1 = Act(self._buf(f"g.e{i}.a1", (M, c)), c)
            fused = self._block_bwd(f"g.e{i}.2", b2, da2, y2, a1, B, h, w, M, da1, next_bn=(b1, y1),
                                    reduce_done=(self.fuse_bn_bwd and i != D - 1) or (i == D - 1 and cur_done),
                                    unpool_src=unpool_src)
            if i > 0:
                d_pool = Act(self._buf(f"g.e{i}.xin", (M, b1.cin)), b1.cin)
                self._block_bwd(f"g.e{i}.1", b1, da1, y1, x_in, B, h, w, M, d_pool, reduce_done=fused)
            else:
                self._block_bwd(f"g.e{i}.1", b1, da1, y1, x_in, B, h, w, M, None, reduce_done=fused)
            g = self._enc_group(i)
            if i == 0 or self._enc_group(i - 1) != g:        # last (shallowest) block of its group
                self._unpack_group(g, on_ready, ranges[g] if ranges else None)
        self._flush_wgrad_group()                     # (nothing left by construction: every group was flushed at its hand-over)
        self._wg_pending = None
        if before_join is not None:
            before_join()                             # the caller's stream has nothing left of the backward pass
        self._join_wgrad()

    # ------------------------------------------------------------------------------------------
    # loss and optimiser
    # ------------------------------------------------------------------------------------------
    def _labels(self, labels):
        if not labels.is_cuda:
            raise hip.HipLibraryError("labels are not on a GPU")
        if labels.dtype not in (torch.int16, torch.int32, torch.int64):
            labels = labels.long()
        return labels.contiguous()

    @_on_device
    def ce_forward(self, logits, labels, class_w, ignore_index=-100, sums=None):
        """Accumulate (sum w*nll, sum w) into ``sums`` (fp64[2], zeroed by the caller or here)."""
        labels = self._labels(labels)
        B, nc, H, W = logits.shape
        if tuple(labels.shape) != (B, H, W):
            raise ValueError(f"labels {tuple(labels.shape)} do not match logits {tuple(logits.shape)}")
        if sums is None:
            sums = torch.zeros(2, dtype=torch.float64, device=logits.device)
        call("crimac_wce_fwd", ptr(logits), ptr(labels), labels.element_size(), ptr(class_w), nc,
             ignore_index, B, H, W, ptr(sums))
        return sums, labels

    @_on_device
    def ce_backward(self, logits, labels, class_w, sums, upstream=1.0, ignore_index=-100):
        B, nc, H, W = logits.shape
        dl = self._buf("g.dlogits", (B, nc, H, W), torch.float32)
        call("crimac_wce_bwd", ptr(logits), ptr(labels), labels.element_size(), ptr(class_w), nc,
             ignore_index, B, H, W, ptr(sums), float(upstream), ptr(dl))
        return dl

    @_on_device
    def sgd_step(self, lr, momentum, grad_scale=1.0, zero_grad=False, guarded=False):
        """guarded: check the flat gradient for inf / NaN first and skip the update if there is one (loss-scaled
        fp16 training); the skip is counted on the device, see ``update_loss_scale``."""
        if guarded:
            if self._scale_state is None:
                self._scale_state = torch.zeros(2, dtype=torch.int32, device=self.device)
            st = self._scale_state
            st[0:1].zero_()
            call("crimac_grad_overflow_flag", ptr(self.flat_g), self.n_flat, ptr(st))
            call("crimac_sgd_momentum_guarded", ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_v), self.n_flat,
                 float(lr), float(momentum), float(grad_scale), 1 if zero_grad else 0, ptr(st))
            # dynamic scale: the pipeline's loss flush re-evaluates it where it synchronises anyway; callers of the bare
            # step (bench.py, engine.train_step, the autograd path + SGDMomentum.step) get the same policy every
            # `loss_scale_check_every` guarded steps -- without it a scale that overflows would skip every step for
            # ever, silently (one host read of two ints per interval)
            self._guarded_steps = getattr(self, "_guarded_steps", 0) + 1
            if self.loss_scale_check_every and self._guarded_steps % self.loss_scale_check_every == 0:
                self.update_loss_scale()
        else:
            call("crimac_sgd_momentum", ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_v), self.n_flat,
                 float(lr), float(momentum), float(grad_scale), 1 if zero_grad else 0)
        self.mark_dirty()

    min_loss_scale = 2.0 ** 4        # floor of the dynamic loss scale (update_loss_scale warns when it is reached)
    loss_scale_check_every = 25      # guarded steps between two host reads of the skip counter inside train_step (0: never)

    def skipped_steps(self):
        """Steps whose update was skipped because a scaled gradient overflowed (host sync)."""
        return 0 if self._scale_state is None else int(self._scale_state[1])

    def update_loss_scale(self, growth_interval=4):
        """Dynamic loss scale, evaluated only where the caller synchronises anyway (the pipeline's loss flush):
        halve the scale if a step was skipped since the last call, double it after ``growth_interval`` clean
        calls (up to 2^24).  Returns the scale now in force."""
        if not self.dynamic_loss_scale:
            return self.loss_scale
        n = self.skipped_steps()
        if n > self._skipped_seen:
            self._skipped_seen, self._good_checks = n, 0
            if self.loss_scale / 2.0 < self.min_loss_scale:
                # a scale this low means the gradients overflow fp16 however they are scaled (diverged weights, inf / NaN
                # in the input): say so once; the guarded step keeps skipping such steps, the scale stays at the floor and
                # grows again from there once steps succeed
                if not getattr(self, "_warned_scale_floor", False):
                    import warnings
                    warnings.warn(f"loss scale reached its floor ({self.min_loss_scale:g}) after {n} skipped steps: the "
                                  "gradients overflow fp16 storage whatever the scale (diverging training or non-finite "
                                  "inputs?)", RuntimeWarning, stacklevel=2)
                    self._warned_scale_floor = True
                self.loss_scale = self.min_loss_scale
            else:
                self.loss_scale = self.loss_scale / 2.0
        else:
            self._good_checks += 1
            if self._good_checks >= growth_interval and self.loss_scale < 2.0 ** 24:
                self.loss_scale *= 2.0
                self._good_checks = 0
        return self.loss_scale

    @_on_device
    def train_step(self, x, labels, class_w, lr, momentum, grad_sync=None, ignore_index=-100, meta=None):
        """Fused step: forward + weighted CE + backward (+ gradient exchange) + SGD.

        Mirrors the loop body of SegPipe.train_model (pipeline.py:163-178) without autograd and
        without a host sync; returns the loss as a 0-d device tensor.
        ``grad_sync(flat_grad) -> scale`` may all-reduce the flat gradient in place.
        """
        logits = self.forward(x, training=True, meta=meta)
        return self._loss_backward_update(logits, labels, class_w, lr, momentum, grad_sync, ignore_index)

    def _loss_backward_update(self, logits, labels, class_w, lr, momentum, grad_sync, ignore_index):
        sums = self.stat[0:2]
        sums, labels = self.ce_forward(logits, labels, class_w, ignore_index, sums=sums)
        ls = float(self.loss_scale)
        # (sum w*nll, sum w) of this rank's batch: the pipeline adds them up over ranks for the logged loss
        self.last_loss_sums = sums.clone()
        dl = self.ce_backward(logits, labels, class_w, sums, ls, ignore_index)
        scale = 1.0
        single = grad_sync is None or (hasattr(grad_sync, "world") and grad_sync.world() == 1
                                       and not getattr(grad_sync, "force", False))

        def finish_exchange():
            """grad_sync.finish() bracketed by the exchange probe's events."""
            if self.exchange_probe is None:
                return grad_sync.finish()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sc = grad_sync.finish()
            e1.record()
            self.exchange_probe.append((e0, e1))
            return sc

        if self.dynamic_loss_scale or ls != 1.0:
            # loss-scaled step (selected by the MODE, not by the current value of the scale: fp16 / plane-pair gradient
            # storage needs the overflow guard whatever the scale has decayed to): all ranges are applied together
            # behind ONE overflow check (a step is applied
            # whole or not at all), so the per-range early updates of the unscaled path are not used
            if grad_sync is not None and hasattr(grad_sync, "launch"):
                self.backward(dl, on_ready=None if single else (lambda lo, hi: grad_sync.launch(self.flat_g, lo, hi)))
                scale = finish_exchange()
            else:
                self.backward(dl)
                if grad_sync is not None:
                    scale = grad_sync(self.flat_g)
            self.sgd_step(lr, momentum, grad_scale=scale / ls, guarded=True)
            return (sums[0] / sums[1]).float()
        if single and self.early_sgd and self._side is not None and len(self._side) == 1 and self.unpack_on_side:
            self._sgd_left = None
            self.mark_dirty()                         # (before the early re-packs register themselves)
            self.backward(dl, before_join=lambda: self._early_sgd(lr, momentum))
            for lo, hi in (self._sgd_left if self._sgd_left is not None else [(0, self.n_flat)]):
                self._sgd_range(lo, hi, lr, momentum)
            return (sums[0] / sums[1]).float()
        if grad_sync is not None and hasattr(grad_sync, "launch"):
            self.backward(dl, on_ready=None if single else (lambda lo, hi: grad_sync.launch(self.flat_g, lo, hi)))
            if not single and self.early_sgd and self.early_sgd_multi and hasattr(grad_sync, "finish_range"):
                # N > 1: the gradient ranges were handed to the exchange in the order the backward pass completed them
                # (decoder first, shallow encoder blocks last).  Each range is applied, and its weight planes re-packed
                # for the next step, as soon as ITS collectives are done -- the update of the decoder (40 % of the
                # parameters) runs while the later ranges are still on the links, instead of the whole optimiser step
                # and the whole re-pack queuing behind the last collective.
                self.mark_dirty()
                probe = self.exchange_probe is not None
                if probe:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                rngs = self.grad_ranges()
                for g, (lo, hi) in enumerate(rngs):
                    scale = grad_sync.finish_range(lo, hi)
                    if probe and g == len(rngs) - 1:
                        e1.record()                   # (the last range's collectives have been waited for)
                        self.exchange_probe.append((e0, e1))
                    self._sgd_range(lo, hi, lr, momentum, scale)
                    if self.early_pack:
                        self._pack_group(g)
                        self._packed_groups.add(g)
                grad_sync.finish()
                return (sums[0] / sums[1]).float()
            scale = finish_exchange()
        else:
            self.backward(dl)
            if grad_sync is not None:
                scale = grad_sync(self.flat_g)
        self.sgd_step(lr, momentum, grad_scale=scale)
        return (sums[0] / sums[1]).float()

    def _early_sgd(self, lr, momentum):
        """Single GPU: while the side stream still runs the last (small) weight gradients, the caller's stream
        is idle -- apply the gradient ranges that are already final there.  Returns the ranges left over."""
        ranges = self.grad_ranges()
        left = []
        for g, (lo, hi) in enumerate(ranges):
            ev = self._unpacked.get(g)
            if ev is None or g == len(ranges) - 1:
                left.append((lo, hi))
                continue
            torch.cuda.current_stream().wait_event(ev)
            self._sgd_range(lo, hi, lr, momentum)
            if self.early_pack:
                self._pack_group(g)                   # its weight planes for the next step, while this stream is idle
                self._packed_groups.add(g)
        self._sgd_left = left

    def _sgd_range(self, lo, hi, lr, momentum, grad_scale=1.0):
        call("crimac_sgd_momentum", ptr(self.flat_p, lo), ptr(self.flat_g, lo), ptr(self.flat_v, lo), hi - lo,
             float(lr), float(momentum), float(grad_scale), 0)

    # ------------------------------------------------------------------------------------------
    # on-GPU augmentation (BASELINE configs[4]; reference: batch/data_augmentation/*)
    # ------------------------------------------------------------------------------------------
    @_on_device
    def augment_batch(self, data_linear, labels, seed, do_noise=True, do_flip=True, refine_labels=None, db_scaled=False):
        """add_noise + flip_x_axis + remove_nan_inf + db_with_limits + NCHW->NHWC in one kernel.
        ``db_scaled``: db_with_limits_scaled instead (1 + dB / 75), the data transform of the metadata configurations
        (batch/transforms.py:50-51).

        data_linear [B,C,H,W] fp32 LINEAR sv on the GPU, labels [B,H,W] int16/32/64 (or None).
        refine_labels=(thr_channel, thr_lo, thr_hi): ``labels`` are RAW annotation ids and the reference's
        training label transform (refine_label_boundary + convert_label_indexing, batch/transforms.py:71-78)
        runs on the GPU between the augmentation and the NaN rule, as in batch/dataset.py:89-103.
        Returns (x_nhwc [B*H*W,16] in the engine's storage type, labels int16 [B,H,W])."""
        self.bind()
        if not data_linear.is_cuda:
            raise hip.HipLibraryError("augment_batch: data is not on a GPU")
        data_linear = data_linear.contiguous().float()
        B, C, H, W = data_linear.shape
        if C != self.in_channels:
            raise ValueError(f"expected {self.in_channels} channels, got {C}")
        x = self._buf("x_nhwc", (B * H * W, CIN_PAD))
        lab_out = self._buf("aug.labels", (B, H, W), torch.int16)
        lab_in = None if labels is None else self._labels(labels)
        aux, thr_c, lo, hi = None, 0, 0.0, 0.0
        if refine_labels is not None:
            if lab_in is None:
                raise ValueError("refine_labels needs labels")
            thr_c, lo, hi = refine_labels
            aux = self._buf("aug.aux", (B, H, W), torch.uint8)
        call("crimac_augment_db_nhwc", self.prec, ptr(data_linear), ptr(lab_in),
             lab_in.element_size() if lab_in is not None else 0, ptr(x), ptr(lab_out), ptr(aux), int(thr_c),
             float(lo), float(hi), B, C, H, W, CIN_PAD, int(seed) & 0xFFFFFFFFFFFFFFFF, 1 if do_noise else 0,
             1 if do_flip else 0, 1 if db_scaled else 0)
        if aux is not None:
            refined = self._buf("aug.labels_refined", (B, H, W), torch.int16)
            call("crimac_refine_labels", ptr(lab_out), 2, ptr(aux), None, int(thr_c), float(lo), float(hi), 1,
                 ptr(refined), B, C, H, W)
            lab_out = refined
        return x, lab_out

    @_on_device
    def flip_planes(self, planes, seed, do_flip=True):
        """The metadata planes [B,Cm,H,W] of a batch flipped along the ping axis under the same per-sample draws as
        ``augment_batch(..., seed)`` flips data and labels (flip_x_axis_metadata, flip_x_axis.py:27-32)."""
        planes = planes.contiguous().float()
        B, Cm, H, W = planes.shape
        out = self._buf("aug.meta", (B, Cm, H, W), torch.float32)
        call("crimac_augment_flip_planes", ptr(planes), ptr(out), B, Cm, H, W, int(seed) & 0xFFFFFFFFFFFFFFFF,
             1 if do_flip else 0)
        return out

    @_on_device
    def train_step_augmented(self, data_linear, labels, class_w, lr, momentum, seed, grad_sync=None,
                             do_noise=True, do_flip=True, ignore_index=-100, refine_labels=None, meta=None):
        """Training step on RAW linear-sv crops: augmentation and dB transform (and, with ``refine_labels``,
        the label transform on raw annotation ids) run on the GPU.  ``meta`` [B,Cm,H,W] (UNet_LateMetInject): the
        reference's *_metadata augmentations (batch/transforms.py:41-42) -- noise on the data planes only, the flip on
        data, metadata and labels alike -- and the scaled dB transform of the metadata configurations (:50-51)."""
        B, _, H, W = data_linear.shape
        if self.lmi:
            meta = self.flip_planes(self._meta(meta, data_linear), seed, do_flip)
        elif meta is not None:
            raise ValueError("this model takes no metadata tensor (late_meta_inject=False)")
        x, lab = self.augment_batch(data_linear, labels, seed, do_noise, do_flip, refine_labels, db_scaled=self.lmi)
        logits = self.forward_nhwc(x, B, H, W, training=True, meta=meta)
        return self._loss_backward_update(logits, lab, class_w, lr, momentum, grad_sync, ignore_index)
