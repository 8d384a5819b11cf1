"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference U-Net hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package must never route through it.

This is a plain-PyTorch (CPU, fp32) functional restatement of the arithmetic that
CRIMAC-classifiers-unet performs on its hot path.  The reference executes every FLOP with stock
torch modules; the third-party dependency that holds the arithmetic is therefore PyTorch itself
(reference pins torch==1.7.1, crimac_unet/requirements.txt:47; golden vectors here were captured
with torch 2.10.0 CPU).  Each function cites the reference lines it restates.

Parity pin: the reference ships no tests or golden vectors, so this oracle is
pinned against outputs of the imported reference itself, generated in the build container by
``tools/make_golden.py`` and committed under ``tests/golden/`` (see ``tests/test_oracle_golden.py``).

The network is held as a flat ``dict`` keyed exactly like ``UNet_Baseline.state_dict()`` so weights
can be exchanged with reference checkpoints (pipeline.py:109-130, :199-203).
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default, unet.py:78
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default
CE_CLASS_WEIGHTS = (10.0, 300.0, 250.0)  # pipeline.py:135
IGNORE_INDEX = -100  # torch default == LABEL_IGNORE_VAL, constants.py:25


def _depth_of(state) -> int:
    d = 0
    while f"down_convs.{d}.main.0.weight" in state:
        d += 1
    return d


def _conv_bn_relu(x, state, conv_key, bn_key, training, new_stats):
    """conv3x3(pad 1, bias) -> BatchNorm2d -> ReLU  (unet.py:35-44, :76-83, :135-136)."""
    y = F.conv2d(x, state[conv_key + ".weight"], state[conv_key + ".bias"], stride=1, padding=1)
    g, b = state[bn_key + ".weight"], state[bn_key + ".bias"]
    if training:
        # batch statistics: biased variance for normalisation, unbiased for the running estimate
        mean = y.mean(dim=(0, 2, 3))
        var_b = y.var(dim=(0, 2, 3), unbiased=False)
        n = y.numel() // y.shape[1]
        with torch.no_grad():
            new_stats[bn_key + ".running_mean"] = (
                (1 - BN_MOMENTUM) * state[bn_key + ".running_mean"] + BN_MOMENTUM * mean.detach())
            new_stats[bn_key + ".running_var"] = (
                (1 - BN_MOMENTUM) * state[bn_key + ".running_var"]
                + BN_MOMENTUM * var_b.detach() * (n / max(n - 1, 1)))
            new_stats[bn_key + ".num_batches_tracked"] = state[bn_key + ".num_batches_tracked"] + 1
    else:
        mean, var_b = state[bn_key + ".running_mean"], state[bn_key + ".running_var"]
    inv = torch.rsqrt(var_b + BN_EPS)
    z = (y - mean[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]
    return torch.relu(z)


def meta_post_processing(state, meta):
    """MetaPostProcessing.forward (unet.py:160-166): Linear(Cm,32)/ReLU/Linear(32,32)/ReLU/Linear(32,1) over the
    channel axis of [N, C, H, W]."""
    p = "post_processing_weights.main."
    h = meta.permute(0, 2, 3, 1)
    h = torch.relu(F.linear(h, state[p + "0.weight"], state[p + "0.bias"]))
    h = torch.relu(F.linear(h, state[p + "2.weight"], state[p + "2.bias"]))
    h = F.linear(h, state[p + "4.weight"], state[p + "4.bias"])
    return h.permute(0, 3, 1, 2)


def unet_forward(state, x, training=False, taps=None, meta=None):
    """``UNet_Baseline.forward`` (unet.py:327-343) on a state dict.

    Returns ``(logits, new_stats)``; ``new_stats`` holds the BatchNorm buffers after the step when
    ``training`` (unet.py:78,81,121-122 -- torch BatchNorm2d train-mode semantics).
    ``taps`` (optional dict) receives intermediate activations for per-layer parity checks.
    """
    depth = _depth_of(state)
    new_stats = OrderedDict()
    skips = []
    for i in range(depth):
        p = f"down_convs.{i}.main."
        x = _conv_bn_relu(x, state, p + "0", p + "1", training, new_stats)
        x = _conv_bn_relu(x, state, p + "3", p + "4", training, new_stats)
        skips.append(x)                       # before_pool, unet.py:90
        if taps is not None:
            taps[f"enc{i}"] = x
        if i < depth - 1:
            x = F.max_pool2d(x, kernel_size=2, stride=2)   # unet.py:85-86
    for i in range(depth - 1):
        p = f"up_convs.{i}."
        skip = skips[-(i + 2)]                # unet.py:336
        up = F.conv_transpose2d(x, state[p + "upconv.weight"], state[p + "upconv.bias"], stride=2)
        if taps is not None:
            taps[f"up{i}"] = up
        x = torch.cat((up, skip), dim=1)      # (from_up, from_down), unet.py:132
        x = _conv_bn_relu(x, state, p + "conv1", p + "bn1", training, new_stats)
        x = _conv_bn_relu(x, state, p + "conv2", p + "bn2", training, new_stats)
        if taps is not None:
            taps[f"dec{i}"] = x
    if meta is not None:      # UNet_LateMetInject.forward (unet.py:386-388): metadata plane behind the decoder output
        x = torch.cat((x, meta_post_processing(state, meta)), dim=1)
    logits = F.conv2d(x, state["conv_final.weight"], state["conv_final.bias"])   # unet.py:342
    return logits, new_stats


def weighted_cross_entropy(logits, labels, class_weights=CE_CLASS_WEIGHTS, ignore_index=IGNORE_INDEX):
    """nn.CrossEntropyLoss(weight=[10,300,250]) (pipeline.py:132-141, call :176).

    sum_p w[y_p] * (-log softmax(z_p)[y_p]) / sum_p w[y_p] over pixels with y_p != ignore_index;
    NaN (0/0) when every pixel is ignored.  Written out by hand so the oracle does
    not simply re-dispatch the op under test.
    """
    w = torch.as_tensor(class_weights, dtype=logits.dtype)
    lab = labels.long()
    valid = lab != ignore_index
    safe = torch.where(valid, lab, torch.zeros_like(lab))
    lse = torch.logsumexp(logits, dim=1)
    picked = torch.gather(logits, 1, safe[:, None]).squeeze(1)
    wy = torch.where(valid, w[safe], torch.zeros((), dtype=logits.dtype))
    return (wy * (lse - picked)).sum() / wy.sum()


def trainable_keys(state):
    """Keys that ``model.parameters()`` yields (everything but BatchNorm buffers)."""
    return [k for k in state
            if not (k.endswith("running_mean") or k.endswith("running_var")
                    or k.endswith("num_batches_tracked"))]


def loss_and_grads(state, x, labels, meta=None):
    """One forward + weighted CE + backward in train mode (pipeline.py:167-177).

    Returns ``(loss, logits, grads, new_stats)`` with ``grads`` keyed like ``state``.
    """
    work = OrderedDict()
    for k, v in state.items():
        work[k] = v.detach().clone()
    for k in trainable_keys(work):
        work[k].requires_grad_(True)
    logits, new_stats = unet_forward(work, x, training=True, meta=meta)
    loss = weighted_cross_entropy(logits, labels)
    keys = trainable_keys(work)
    gs = torch.autograd.grad(loss, [work[k] for k in keys])
    return loss.detach(), logits.detach(), OrderedDict(zip(keys, gs)), new_stats


def sgd_momentum_step(state, grads, velocity, lr, momentum):
    """optim.SGD(lr, momentum) step (pipeline.py:156, :178): v<-mu*v+g (v0=g); p<-p-lr*v."""
    for k, g in grads.items():
        if k not in velocity:
            velocity[k] = g.clone()
        else:
            velocity[k].mul_(momentum).add_(g)
        state[k] = state[k] - lr * velocity[k]
    return state, velocity


def train_steps(state, batches, lr, momentum, lr_reduction=1.0, lr_step=10 ** 9):
    """The loop body of ``SegPipe.train_model`` (pipeline.py:161-193) for a list of (x, labels)."""
    state = OrderedDict((k, v.clone()) for k, v in state.items())
    velocity, losses = {}, []
    cur_lr = lr
    for it, (x, lab) in enumerate(batches):
        loss, _, grads, new_stats = loss_and_grads(state, x, lab)
        state, velocity = sgd_momentum_step(state, grads, velocity, cur_lr, momentum)
        state.update(new_stats)
        losses.append(float(loss))
        if (it + 1) % lr_step == 0:          # ExponentialLR, pipeline.py:188-189
            cur_lr *= lr_reduction
    return state, losses


def predict(state, x, return_softmax=False, meta=None):
    """``SegPipe.predict_batch`` (pipeline.py:205-219): eval-mode forward (+ softmax over classes)."""
    with torch.no_grad():
        logits, _ = unet_forward(state, x.float(), training=False, meta=None if meta is None else meta.float())
        return F.softmax(logits, dim=1) if return_softmax else logits


def set_label_ignore_val(labels):
    """pipeline.py:222-239: {-70,-30,-100,-10} -> -100 ; -50 (below seabed) -> 0."""
    out = labels.clone()
    for v in (-70, -30, -100, -10):
        out[labels == v] = IGNORE_INDEX
    out[labels == -50] = 0
    return out
