"""Loss and optimiser objects with the torch surface the reference's training loop uses, over HIP.

* :class:`WeightedCrossEntropy` stands in for ``nn.CrossEntropyLoss(weight=...)`` as returned by
  ``SegPipe.get_criterion`` (pipeline.py:132-141; called at :176, :264);
* :class:`SGDMomentum` for ``optim.SGD(model.parameters(), lr, momentum)`` (pipeline.py:156, :178);
* :class:`ExponentialLR` for ``optim.lr_scheduler.ExponentialLR`` (pipeline.py:157, :188-189).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip
from .hip import call, ptr


def _prep_labels(labels):
    if not labels.is_cuda:
        raise hip.HipLibraryError("labels are not on a GPU: the HIP path has no CPU fallback")
    if labels.dtype not in (torch.int16, torch.int32, torch.int64):
        labels = labels.long()
    return labels.contiguous()


class _WCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight, ignore_index):
        logits = logits.contiguous().float()
        labels = _prep_labels(labels)
        B, nc, H, W = logits.shape
        if tuple(labels.shape) != (B, H, W):
            raise ValueError(f"labels {tuple(labels.shape)} do not match logits {tuple(logits.shape)}")
        sums = torch.zeros(2, dtype=torch.float64, device=logits.device)
        call("crimac_wce_fwd", ptr(logits), ptr(labels), labels.element_size(), ptr(weight), nc,
             ignore_index, B, H, W, ptr(sums))
        ctx.save_for_backward(logits, labels, weight, sums)
        ctx.ignore_index = ignore_index
        return (sums[0] / sums[1]).float()      # NaN if every pixel is ignored, as in torch

    @staticmethod
    def backward(ctx, g):
        logits, labels, weight, sums = ctx.saved_tensors
        B, nc, H, W = logits.shape
        dl = torch.empty_like(logits)
        # upstream is folded in afterwards so no host sync is needed to read it
        call("crimac_wce_bwd", ptr(logits), ptr(labels), labels.element_size(), ptr(weight), nc,
             ctx.ignore_index, B, H, W, ptr(sums), 1.0, ptr(dl))
        return dl * g, None, None, None


class WeightedCrossEntropy(nn.Module):
    """Per-pixel class-weighted cross entropy, mean over non-ignored pixels weighted by w[y]."""

    def __init__(self, weight, ignore_index=-100):
        super().__init__()
        self.register_buffer("weight", torch.as_tensor(weight, dtype=torch.float32))
        self.ignore_index = ignore_index

    def forward(self, logits, labels):
        if logits.dim() != 4:
            raise ValueError("expected logits [B, n_classes, H, W]")
        if self.weight.device != logits.device:
            self.weight = self.weight.to(logits.device)
        return _WCEFunction.apply(logits, labels, self.weight, self.ignore_index)


class SGDMomentum:
    """SGD with momentum (no dampening / nesterov / weight decay) as one flat multi-tensor kernel."""

    def __init__(self, model, lr, momentum=0.0):
        self.model = model
        self.param_groups = [{"lr": float(lr), "momentum": float(momentum),
                              "params": list(model.parameters())}]

    @property
    def engine(self):
        return self.model.engine

    def zero_grad(self, set_to_none=False):
        eng = self.engine
        eng.bind()
        eng.flat_g.zero_()

    def step(self):
        eng = self.engine
        eng.bind()
        g = self.param_groups[0]
        # loss-scaled precision (fp16): a step whose gradients overflowed is skipped, as torch.cuda.amp.GradScaler does
        eng.sgd_step(g["lr"], g["momentum"], guarded=eng.dynamic_loss_scale or eng.loss_scale != 1.0)

    def state_dict(self):
        g = self.param_groups[0]
        return {"lr": g["lr"], "momentum": g["momentum"], "velocity": self.engine.flat_v.clone()}


class ExponentialLR:
    def __init__(self, optimizer, gamma):
        self.optimizer, self.gamma = optimizer, float(gamma)

    def step(self):
        for g in self.optimizer.param_groups:
            g["lr"] *= self.gamma

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]
