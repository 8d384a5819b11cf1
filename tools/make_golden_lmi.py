#!/usr/bin/env python3
"""Golden fixture for late metadata injection: runs the *imported reference* UNet_LateMetInject
(crimac_unet/models/unet.py:346-391) in the build container and stores inputs-free outputs under
tests/golden/lmi.npz (weights, crops and metadata planes are regenerated from seeds by
crimac_classifiers_unet_amd.synth).  Also cross-checks oracle/unet_oracle.py's metadata branch.

Usage: python tools/make_golden_lmi.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/crimac_unet")

from crimac_classifiers_unet_amd import synth  # noqa: E402
from oracle import unet_oracle as orc  # noqa: E402
import models.unet as ref_models  # noqa: E402  (the reference)

torch.set_num_threads(8)
CM, HW = 7, 64


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    sd = synth.synth_state_dict(seed=0, meta_in_channels=CM)
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, HW, HW, seed=1))
    meta = torch.from_numpy(synth.synth_metadata(2, CM, HW, HW, seed=3))
    lab = torch.from_numpy(synth.synth_labels(2, HW, HW, seed=2))
    net = ref_models.UNet_LateMetInject(n_classes=3, in_channels=4, meta_in_channels=CM)
    assert list(net.state_dict().keys()) == list(sd.keys()), "state_dict key order differs from the reference"
    assert all(tuple(v.shape) == tuple(sd[k].shape) or v.numel() == 1 for k, v in net.state_dict().items())
    net.load_state_dict(sd)
    net.eval()
    with torch.no_grad():
        logits_eval = net(x, meta)
    assert rel(orc.predict(sd, x, meta=meta), logits_eval) < 2e-6
    net.train()
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([10.0, 300, 250]))
    out = net(x, meta)
    loss = crit(out, lab.long())
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    o_loss, o_logits, o_grads, _ = orc.loss_and_grads(sd, x, lab, meta=meta)
    print("oracle vs reference: train logits", rel(o_logits, out.detach()), "loss", float(o_loss), float(loss))
    assert rel(o_logits, out.detach()) < 1e-5 and abs(float(o_loss) - float(loss)) < 1e-6 * abs(float(loss))
    fix = {"logits_eval": logits_eval.numpy(), "logits_train": out.detach().numpy(), "loss": np.float64(float(loss)),
           "keys": np.array(list(sd.keys()))}
    for k, g in grads.items():
        fix["gnorm/" + k] = np.float64(float(g.double().norm()))
        if k.startswith("post_processing_weights") or k.startswith("conv_final"):
            fix["grad/" + k] = g.numpy()
            assert rel(o_grads[k], g) < 1e-3, k
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lmi.npz"), **fix)
    print("wrote tests/golden/lmi.npz", sum(v.nbytes for v in fix.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
