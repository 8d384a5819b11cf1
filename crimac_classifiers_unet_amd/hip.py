"""ctypes binding of libcrimac_unet_hip.so (the C ABI declared in include/crimac_unet_hip.h).

PyTorch is used only for device memory and streams: tensors are passed as raw ``data_ptr()``s plus
the current ``torch.cuda`` stream handle.  There is NO CPU fallback: if the library is missing or a
tensor is not on a GPU, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import build as _build

PREC_BF16 = 0
PREC_F32X3 = 1
PREC_F32X6 = 2
PREC_FP16 = 3
PREC_F32H3 = 4
PREC_H3P = 5             # pre-split fp16 plane pairs (CRIMAC_PREC_H3P): F32H3's arithmetic on the LDS-DMA kernels
PREC_H3F_BWD = 6         # CRIMAC_PREC_H3F_BWD: the backward pass of 'h3f' (fp16 MFMA operands, everything else as H3P)
# 'h3f': the forward pass of 'h3p' (same kernels, same bits) with fp16 MFMA operands in the BACKWARD pass -- an engine-level
# mode (engine.py): forward launches carry PREC_H3P, the backward pass's contractions and BatchNorm-backward PREC_H3F_BWD
PREC_NAMES = {"bf16": PREC_BF16, "f32x3": PREC_F32X3, "f32x6": PREC_F32X6, "fp16": PREC_FP16, "f32h3": PREC_F32H3,
              "h3p": PREC_H3P, "h3f": PREC_H3P}
MFMAS_PER_PRODUCT = {PREC_BF16: 1, PREC_FP16: 1, PREC_F32X3: 3, PREC_F32H3: 3, PREC_H3P: 3, PREC_F32X6: 6, PREC_H3F_BWD: 1}
PREC_PLANES = {PREC_BF16: 1, PREC_F32X3: 2, PREC_F32X6: 3, PREC_FP16: 1, PREC_F32H3: 2, PREC_H3P: 2}   # 16-bit planes per operand
# `planes` argument of the packing entry points (CRIMAC_PLANES_* of the header): bits 0-3 planes, bit 4 / 5 forward /
# input-gradient planes in IEEE half, bits 8-15 log2 of the scale on the forward planes
PLANES_FP16 = 1 | 16 | 32
PLANES_F32H3 = 2 | 16 | (8 << 8)
PLANES_FWD_FRAG = 65536       # crimac_pack_conv3x3: the forward plane fragment-major (CRIMAC_PLANES_FWD_FRAG)
PLANES_INTERLEAVED = 64      # both planes in the `hi` buffer, [32 hi | 32 lo] per 32-channel block of a row
PLANES_H3P = 2 | 16 | 32 | PLANES_INTERLEAVED | 128 | (8 << 8)        # CRIMAC_PLANES_H3P
PREC_PLANES_ARG = {PREC_BF16: 1, PREC_F32X3: 2, PREC_F32X6: 3, PREC_FP16: PLANES_FP16, PREC_F32H3: PLANES_F32H3,
                   PREC_H3P: PLANES_H3P}
EPI_RELU, EPI_OUT_PLANES, EPI_CIN4 = 1, 2, 4      # `relu` argument of the convolution entry points (CRIMAC_EPI_*)
EPI_WFRAG = 16                # ... the weight plane is fragment-major (CRIMAC_EPI_WFRAG)
EPI_WROWS = 32                # ... and the channel-split kernel's rows form reads it (CRIMAC_EPI_WROWS)
LAYER_FWD_FRAG, LAYER_DG_FRAG = 16, 32      # crimac_layer_desc.kind flags (CRIMAC_LAYER_*_FRAG)
# precision the BACKWARD kernels (input gradients, weight gradients) are called with: F32H3 is a forward-operand
# mode (fp16 planes have no range for gradients), its backward pass runs on the 2-plane bf16 split
PREC_BACKWARD = {PREC_F32H3: PREC_F32X3}
PREC_16BIT = (PREC_BF16, PREC_FP16)

ABI_VERSION = 6          # CRIMAC_ABI_VERSION of include/crimac_unet_hip.h this binding was written against

_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float

# name -> argtypes, exactly the prototypes of include/crimac_unet_hip.h
SIGNATURES = {
    "crimac_igemm_conv": [_i, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i,
                          _vp, _l, _i, _i, _i, _vp],
    "crimac_conv3x3": [_i, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _i,
                       _vp, _l, _vp, _l, _vp],
    "crimac_conv3x3_pool": [_i, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _l, _i, _vp, _l, _vp],
    "crimac_conv3x3_cols": [_i, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _i,
                            _vp, _l, _vp, _l, _i, _i, _vp],
    "crimac_upconv2x2_dgrad_bnb": [_vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _i, _vp],
    "crimac_upconv2x2_dgrad_bnb_prec": [_i, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _i,
                                        _vp],
    "crimac_sum_replicas": [_vp, _i, _l, _i, _vp, _vp, _vp, _vp, _vp],
    "crimac_wgrad": [_i, _i, _vp, _l, _i, _vp, _l, _i, _i, _i, _i, _vp, _i, _vp],
    "crimac_wgrad_partials": [_i, _i, _vp, _l, _i, _vp, _l, _i, _i, _i, _i, _vp, _l, _i, _vp],
    "crimac_wgrad_group": [_i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp],
    "crimac_pack_conv3x3": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "crimac_pack_upconv2x2": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "crimac_unpack_wgrad_conv3x3": [_vp, _i, _i, _i, _vp, _vp],
    "crimac_unpack_wgrad_upconv2x2": [_vp, _i, _i, _vp, _vp],
    "crimac_pack_layers": [_vp, _i, _i, _vp],
    "crimac_unpack_wgrad_layers": [_vp, _i, _vp],
    "crimac_nchw_to_nhwc": [_i, _vp, _vp, _i, _i, _i, _i, _l, _vp],
    "crimac_colstats": [_i, _vp, _l, _l, _i, _vp, _vp, _vp],
    "crimac_colsum_f32": [_i, _vp, _l, _l, _i, _vp, _vp],
    "crimac_bn_finalize": [_vp, _vp, _i, _l, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "crimac_bn_act_pool": [_i, _vp, _l, _vp, _vp, _i, _vp, _l, _vp, _l, _i, _i, _i, _i, _vp],
    "crimac_bn_train_act_pool": [_i, _vp, _l, _vp, _vp, _i, _l, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _l, _i, _vp, _l,
                                 _vp, _l, _i, _i, _i, _i, _vp],
    "crimac_unpool_add": [_i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _i,
                          _vp],
    "crimac_bn_bwd_reduce": [_i, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _l, _i, _vp, _vp, _vp],
    "crimac_bn_bwd_apply": [_i, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _i, _vp, _l, _vp,
                            _vp, _vp, _vp],
    "crimac_bn_bwd_apply_replicas": [_i, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _i, _l, _l, _i, _vp, _l, _vp, _vp, _vp],
    "crimac_unpool_bn_bwd_apply_replicas": [_i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _i, _l, _vp, _l, _i, _i, _i, _i,
                                            _vp, _vp, _vp],
    "crimac_head_fwd": [_i, _vp, _l, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "crimac_head_bwd": [_i, _vp, _vp, _l, _i, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp,
                        _i, _vp],
    "crimac_softmax_nchw": [_vp, _vp, _i, _i, _i, _i, _vp],
    "crimac_wce_fwd": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "crimac_wce_bwd": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _f, _vp, _vp],
    "crimac_sgd_momentum": [_vp, _vp, _vp, _l, _f, _f, _f, _i, _vp],
    "crimac_grad_overflow_flag": [_vp, _l, _vp, _vp],
    "crimac_sgd_momentum_guarded": [_vp, _vp, _vp, _l, _f, _f, _f, _i, _vp, _vp],
    "crimac_meta_planes": [_vp, _i, _i, _i, _i, C.c_double, _vp, _i, _vp, _i, _vp, _i, _vp, _vp],
    "crimac_meta_mlp_fwd": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "crimac_meta_inject_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "crimac_meta_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                        _vp, _vp],
    "crimac_gather_patches": [_i, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _l, _vp],
    "crimac_gather_patches_memm": [_i, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _l, _vp, _vp],
    "crimac_scatter_patches_ex": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i,
                                  _i, _i, _vp, _i, _vp],
    "crimac_augment_db_nhwc": [_i, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _i, _i, _i, _i, _l, C.c_ulonglong,
                               _i, _i, _i, _vp],
    "crimac_augment_flip_planes": [_vp, _vp, _i, _i, _i, _i, C.c_ulonglong, _i, _vp],
    "crimac_refine_labels": [_vp, _i, _vp, _vp, _i, _f, _f, _i, _vp, _i, _i, _i, _i, _vp],
    "crimac_pr_histogram": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "crimac_mfma_calibrate": [_i, _i, _vp, _vp],
    "crimac_labels_test_transform": [_vp, _i, _vp, _i, _f, _f, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i,
                                     _i, _i],
    "crimac_labels_extend_mask": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i],
    "crimac_scatter_patches": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i,
                               _i, _vp, _vp],
}



class LayerDesc(C.Structure):
    """crimac_layer_desc (include/crimac_unet_hip.h): one Conv2d / ConvTranspose2d layer's buffers."""
    _fields_ = [("w", _vp), ("grad", _vp), ("dw", _vp), ("fwd_hi", _vp), ("fwd_lo", _vp), ("dg_hi", _vp),
                ("dg_lo", _vp), ("kind", _i), ("Co", _i), ("Ci", _i), ("Ci_pad", _i), ("dw_splits", _i),
                ("dw_stride", _l)]


class WgradGroupLayer(C.Structure):
    """crimac_wgrad_group_layer (include/crimac_unet_hip.h): one conv3x3 layer of a grouped weight-gradient launch."""
    _fields_ = [("f", _vp), ("f_ld", _l), ("CF", _i), ("s", _vp), ("s_ld", _l), ("CS", _i), ("Hf", _i), ("Wf", _i),
                ("dw", _vp), ("tiles_y", _i), ("tiles_x", _i), ("ntiles", _l), ("tiles_per_block", _i), ("nsplits", _i)]


WGRAD_GROUP_MAX_LAYERS = 16      # CRIMAC_WGRAD_GROUP_MAX_LAYERS
MASK_PER_PATCH = -2147483648      # CRIMAC_MASK_PER_PATCH


_lib = None


class HipLibraryError(RuntimeError):
    pass


def library_path() -> str:
    return os.environ.get("CRIMAC_LIB") or _build.LIB_PATH       # override: kernel experiments only


def load_library():
    """dlopen the in-tree library and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} is missing: build it with `python -m crimac_classifiers_unet_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the U-Net hot path.")
    lib = C.CDLL(path)
    lib.crimac_version.restype = C.c_int
    lib.crimac_version.argtypes = []
    lib.crimac_last_error.restype = C.c_char_p
    lib.crimac_last_error.argtypes = []
    ver = lib.crimac_version()
    if ver != ABI_VERSION:
        raise HipLibraryError(f"{path} has ABI version {ver}, this binding needs {ABI_VERSION}: rebuild it "
                              "(`python -m crimac_classifiers_unet_amd.build --force`)")
    if not hasattr(lib, "crimac_layer_desc_size"):
        raise HipLibraryError(f"{path} does not export crimac_layer_desc_size: stale build")
    lib.crimac_layer_desc_size.restype = C.c_int
    lib.crimac_layer_desc_size.argtypes = []
    if lib.crimac_layer_desc_size() != C.sizeof(LayerDesc):
        raise HipLibraryError(f"{path}: crimac_layer_desc is {lib.crimac_layer_desc_size()} bytes in the library, "
                              f"{C.sizeof(LayerDesc)} in this binding")
    lib.crimac_wgrad_splits.restype = C.c_int          # (returns a count, not a status; no stream argument)
    lib.crimac_wgrad_splits.argtypes = [_i, _i, _i, _i, _i, _i, _i, _i]
    lib.crimac_wgrad_group_layer_size.restype = C.c_int
    lib.crimac_wgrad_group_layer_size.argtypes = []
    if lib.crimac_wgrad_group_layer_size() != C.sizeof(WgradGroupLayer):
        raise HipLibraryError(f"{path}: crimac_wgrad_group_layer is {lib.crimac_wgrad_group_layer_size()} bytes in the "
                              f"library, {C.sizeof(WgradGroupLayer)} in this binding")
    lib.crimac_wgrad_group_plan.restype = C.c_int      # (host-only planner: returns the queue capacity, no stream)
    lib.crimac_wgrad_group_plan.argtypes = [_i, _vp, _i, _i, _i, _vp, _i, _vp]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = C.c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc: int, name: str):
    if rc != 0:
        msg = load_library().crimac_last_error().decode(errors="replace")
        raise HipLibraryError(f"{name} failed ({rc}): {msg}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, offset_elems: int = 0):
    """Raw device pointer of a tensor (+ element offset); None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("tensor is not on a GPU: the HIP path has no CPU fallback")
    return C.c_void_p(t.data_ptr() + offset_elems * t.element_size())


class Act:
    """A channel slice of an NHWC activation buffer: base tensor + channel offset + pixel stride."""

    __slots__ = ("t", "off", "ld", "C")

    def __init__(self, t, C_, off=0, ld=None):
        self.t = t
        self.off = off
        self.ld = ld if ld is not None else t.shape[-1]
        self.C = C_

    @property
    def p(self):
        return ptr(self.t, self.off)

    def slice(self, off, C_):
        return Act(self.t, C_, self.off + off, self.ld)


# When set to a list (bench.py), every call that passes ``flops=`` is bracketed by HIP events on
# the current stream and (name, flops, start, end) is appended.
PROFILE = None


def call(name: str, *args, flops=None, mfmas=1):
    """``flops``: algorithmic FLOPs of the launch (profiled launches only); ``mfmas``: MFMAs the kernel spends per
    algorithmic product (1 for 16-bit operands, 3 for plane pairs ...): flops * mfmas is what the MFMA pipe executes, the
    figure a roofline against the dense 16-bit peak needs when one step mixes precisions ('h3f')."""
    lib = load_library()
    if PROFILE is not None and flops is not None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _check(getattr(lib, name)(*args, _stream()), name)
        e.record()
        PROFILE.append((name, flops, s, e, mfmas))
        return
    _check(getattr(lib, name)(*args, _stream()), name)
