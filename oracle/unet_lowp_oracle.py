"""ORACLE -- TEST INFRASTRUCTURE ONLY.  The reference arithmetic with 16-bit STORAGE rounding.

Only ``tests/`` may import this module.

``oracle/unet_oracle.py`` restates the reference's fp32 arithmetic.  The throughput precisions of the HIP
path ('bf16', 'fp16') keep fp32 accumulation, statistics, loss and parameters but STORE every activation and
every activation gradient as a 16-bit value and feed the MFMAs 16-bit weights.  Against the fp32 oracle that
shows as a 1e-2 envelope (the reference itself under ``torch.autocast`` sits there), which
cannot catch a wrong-but-finite gradient.  This variant rounds at exactly the points where the engine stores
(crimac_classifiers_unet_amd/engine.py; csrc/conv_epilogue.h, elementwise.hip) and is fp32 everywhere else,
so the only differences left against the HIP path are fp32 summation order and the rare 1-ulp rounding flips
it causes -- the comparison is deterministic and tight on EVERY gradient.

Storage points (T = bf16 or fp16):
  * network input x -> T (crimac_nchw_to_nhwc);
  * conv / transposed-conv weights -> T for the forward and input-gradient products (crimac_pack_layers);
    the weight GRADIENT is an fp32 sum of products of stored T values, not rounded;
  * conv output y = conv(x_T, w_T) + bias -> T; BatchNorm batch statistics are taken from the stored y;
  * activation a = relu(y * scale + shift) -> T (scale = gamma * invstd, shift = beta - mean * scale in fp32);
  * transposed-conv output -> T; max-pool of stored values is exact;
  * 1x1 head: fp32 weights on the stored activation; logits, loss, dlogits fp32;
  * backward: every activation gradient that goes to memory is T: dy (BatchNorm backward output), the
    input gradient of every conv / transposed conv, the sum unpool(d_pool) + d_skip, d(head input).
Reference lines restated: unet.py:35-60, :76-92, :112-136, :327-343; pipeline.py:132-141, :176-177.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

from .unet_oracle import BN_EPS, BN_MOMENTUM, _depth_of, trainable_keys, weighted_cross_entropy

STORAGE = {"bf16": torch.bfloat16, "fp16": torch.float16}


class _StoreBoth(torch.autograd.Function):
    """Value is stored as T on the way forward, its gradient is stored as T on the way back."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).float(), None


def _st(x, dt):
    return _StoreBoth.apply(x, dt)


def _wq(w, dt):
    """T-rounded weight for the products; the gradient goes straight to the fp32 master (no rounding of dW)."""
    return w + (w.detach().to(dt).float() - w.detach())


def _block(x, state, conv_key, bn_key, dt, new_stats):
    """conv3x3 -> BatchNorm2d(train) -> ReLU with the engine's storage rounding."""
    y = _st(F.conv2d(x, _wq(state[conv_key + ".weight"], dt), state[conv_key + ".bias"], padding=1), dt)
    g, b = state[bn_key + ".weight"], state[bn_key + ".bias"]
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    n = y.numel() // y.shape[1]
    with torch.no_grad():
        new_stats[bn_key + ".running_mean"] = (1 - BN_MOMENTUM) * state[bn_key + ".running_mean"] + BN_MOMENTUM * mean
        new_stats[bn_key + ".running_var"] = ((1 - BN_MOMENTUM) * state[bn_key + ".running_var"]
                                              + BN_MOMENTUM * var * (n / max(n - 1, 1)))
    scale = g * torch.rsqrt(var + BN_EPS)
    shift = b - mean * scale
    return _st(torch.relu(y * scale[None, :, None, None] + shift[None, :, None, None]), dt)


def unet_forward_train(state, x, storage="bf16"):
    dt = STORAGE[storage]
    depth = _depth_of(state)
    new_stats = OrderedDict()
    x = x.to(dt).float()
    skips = []
    for i in range(depth):
        p = f"down_convs.{i}.main."
        x = _block(x, state, p + "0", p + "1", dt, new_stats)
        x = _block(x, state, p + "3", p + "4", dt, new_stats)
        skips.append(x)
        if i < depth - 1:
            x = _st(F.max_pool2d(x, 2, 2), dt)          # (the gradient of the pooled tensor is a stored conv dgrad)
    for i in range(depth - 1):
        p = f"up_convs.{i}."
        up = _st(F.conv_transpose2d(x, _wq(state[p + "upconv.weight"], dt), state[p + "upconv.bias"], stride=2), dt)
        x = _st(torch.cat((up, skips[-(i + 2)]), dim=1), dt)      # d(concat) is one stored conv dgrad
        x = _block(x, state, p + "conv1", p + "bn1", dt, new_stats)
        x = _block(x, state, p + "conv2", p + "bn2", dt, new_stats)
    logits = F.conv2d(x, state["conv_final.weight"], state["conv_final.bias"])
    return logits, new_stats


def loss_and_grads(state, x, labels, storage="bf16", loss_scale=1.0):
    """Train-mode forward + weighted CE + backward with 16-bit storage rounding: (loss, logits, grads, stats).

    ``loss_scale``: the engine's fp16 mode multiplies the loss gradient by a power of two so that the stored
    activation gradients sit in fp16's normal range, and divides the parameter gradients by it again
    (engine._loss_backward_update); restated here by differentiating ``loss * loss_scale`` -- every stored
    gradient is then rounded at the same scaled magnitude -- and unscaling the result."""
    work = OrderedDict((k, v.detach().clone()) for k, v in state.items())
    keys = trainable_keys(work)
    for k in keys:
        work[k].requires_grad_(True)
    logits, new_stats = unet_forward_train(work, x, storage)
    loss = weighted_cross_entropy(logits, labels)
    gs = torch.autograd.grad(loss * float(loss_scale), [work[k] for k in keys])
    gs = [g / float(loss_scale) for g in gs]
    return loss.detach(), logits.detach(), OrderedDict(zip(keys, gs)), new_stats


def predict(state, x, storage="bf16", return_softmax=False):
    """Eval forward as the engine runs it: BatchNorm folded into the conv (W*s -> T, bias' = (b-rm)*s+beta in fp32,
    engine._pack_eval / SURVEY.md A4), ReLU in the conv epilogue, every activation stored as T."""
    dt = STORAGE[storage]
    depth = _depth_of(state)

    def block(x, ck, bk):
        s = state[bk + ".weight"] * torch.rsqrt(state[bk + ".running_var"] + BN_EPS)
        w = (state[ck + ".weight"] * s[:, None, None, None]).to(dt).float()
        b = (state[ck + ".bias"] - state[bk + ".running_mean"]) * s + state[bk + ".bias"]
        return torch.relu(F.conv2d(x, w, b, padding=1)).to(dt).float()

    with torch.no_grad():
        x = x.float().to(dt).float()
        skips = []
        for i in range(depth):
            p = f"down_convs.{i}.main."
            x = block(block(x, p + "0", p + "1"), p + "3", p + "4")
            skips.append(x)
            if i < depth - 1:
                x = F.max_pool2d(x, 2, 2)
        for i in range(depth - 1):
            p = f"up_convs.{i}."
            up = F.conv_transpose2d(x, state[p + "upconv.weight"].to(dt).float(), state[p + "upconv.bias"],
                                    stride=2).to(dt).float()
            x = torch.cat((up, skips[-(i + 2)]), dim=1)
            x = block(block(x, p + "conv1", p + "bn1"), p + "conv2", p + "bn2")
        logits = F.conv2d(x, state["conv_final.weight"], state["conv_final.bias"])
        return F.softmax(logits, dim=1) if return_softmax else logits
