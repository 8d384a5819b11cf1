#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by running the *imported reference*.

Runs only in the build container (needs /root/reference; never on the GPU box).  The reference
source is imported, never copied: fixtures hold inputs/outputs only.  Weights and inputs are
regenerated from seeds by ``crimac_classifiers_unet_amd.synth`` so they need not be stored.

Also cross-checks the CPU oracle (oracle/unet_oracle.py) against the reference while it is here.

Usage: python tools/make_golden.py
"""
import os
import re
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/crimac_unet"
sys.path.insert(0, REF)

from crimac_classifiers_unet_amd import synth  # noqa: E402
from oracle import unet_oracle as orc  # noqa: E402

import models.unet as ref_models  # noqa: E402  (the reference)

OUT = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)
torch.set_num_threads(8)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def is_pre_bn_conv_bias(k):
    """Conv biases followed by BatchNorm: gradient is mathematically zero (rounding noise only)."""
    return bool(re.fullmatch(r"down_convs\.\d+\.main\.[03]\.bias|up_convs\.\d+\.conv[12]\.bias", k))


def run_case(tag, start_filts, hw, keep_full_grads):
    sd = synth.synth_state_dict(start_filts=start_filts, seed=0)
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, hw, hw, seed=1))
    lab = torch.from_numpy(synth.synth_labels(2, hw, hw, seed=2))
    net = ref_models.UNet_Baseline(n_classes=3, in_channels=4, depth=5, start_filts=start_filts)
    net.load_state_dict(sd)

    # eval forward (pipeline.py:205-219)
    net.eval()
    with torch.no_grad():
        logits_eval = net(x)
    o_eval = orc.predict(sd, x)
    print(tag, "oracle vs ref eval logits rel", rel(o_eval, logits_eval))
    assert rel(o_eval, logits_eval) < 2e-6

    # train step (pipeline.py:161-178)
    net.train()
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([10.0, 300, 250]))
    opt = torch.optim.SGD(net.parameters(), lr=0.005, momentum=0.95)
    losses, logits_train, grads, stats1 = [], None, None, None
    for it in range(3):
        opt.zero_grad()
        out = net(x)
        loss = crit(out, lab.long())
        loss.backward()
        if it == 0:
            logits_train = out.detach().clone()
            grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
            stats1 = {k: v.detach().clone() for k, v in net.state_dict().items()
                      if "running" in k or "num_batches" in k}
        opt.step()
        losses.append(float(loss))
    final = {k: v.detach().clone() for k, v in net.state_dict().items()}

    # oracle cross-check: grads, stats, 3-step trajectory
    o_loss, o_logits, o_grads, o_stats = orc.loss_and_grads(sd, x, lab)
    print(tag, "oracle train logits rel", rel(o_logits, logits_train), "loss", float(o_loss), losses[0])
    assert rel(o_logits, logits_train) < 5e-6
    assert abs(float(o_loss) - losses[0]) < 1e-5 * abs(losses[0])
    worst, worst_key = 0.0, None
    for k in grads:
        # conv biases feeding a BatchNorm have a mathematically-zero gradient (pure rounding noise)
        if is_pre_bn_conv_bias(k):
            continue
        r = float((o_grads[k] - grads[k]).norm() / grads[k].norm())
        if r > worst:
            worst, worst_key = r, k
    print(tag, "oracle grads worst L2-rel", worst, worst_key)
    # fp32 gradients of this net are only reproducible to ~4e-3 (L2) / ~3e-2 (max): 1e-6 forward
    # differences flip ReLU / max-pool decisions.  Measured below against an fp64 run of the reference.
    assert worst < 1e-2

    # fp64 run of the reference = ground truth for the fp32 noise floor
    net64 = ref_models.UNet_Baseline(n_classes=3, in_channels=4, depth=5, start_filts=start_filts).double()
    net64.load_state_dict(sd)
    net64.train()
    crit64 = torch.nn.CrossEntropyLoss(weight=torch.tensor([10.0, 300, 250], dtype=torch.float64))
    out64 = net64(x.double())
    loss64 = crit64(out64, lab.long())
    loss64.backward()
    grads64 = {k: p.grad.detach() for k, p in net64.named_parameters()}
    for k in stats1:
        assert rel(o_stats[k].float(), stats1[k].float()) < 1e-5, k
    o_state, o_losses = orc.train_steps(sd, [(x, lab)] * 3, lr=0.005, momentum=0.95)
    print(tag, "losses ref", losses, "oracle", o_losses)
    assert np.allclose(o_losses, losses, rtol=2e-4)

    fix = {
        "start_filts": np.int64(start_filts), "hw": np.int64(hw),
        "logits_eval": logits_eval.numpy(), "logits_train": logits_train.numpy(),
        "losses": np.asarray(losses, dtype=np.float64),
    }
    fix["loss64"] = np.float64(loss64.detach())
    for k, g in grads.items():
        fix["gnorm/" + k] = np.float64(g.double().norm())
        fix["gnorm64/" + k] = np.float64(grads64[k].norm())
        # L2-relative distance of the reference's own fp32 gradient from its fp64 gradient
        fix["gnoise/" + k] = np.float64((g.double() - grads64[k]).norm() / grads64[k].norm().clamp_min(1e-300))
        if keep_full_grads or g.numel() <= 4096 or k in (
                "down_convs.0.main.0.weight", "conv_final.weight", "up_convs.3.upconv.weight"):
            fix["grad/" + k] = g.numpy()
            if not keep_full_grads:
                fix["grad64/" + k] = grads64[k].numpy().astype(np.float32)
    for k, v in stats1.items():
        fix["stat1/" + k] = v.numpy()
    for k, v in final.items():
        fix["final_norm/" + k] = np.float64(v.double().norm())
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **fix)
    return sd, x, lab, logits_eval


def run_pipeline_case(sd, x, logits_eval):
    """Exercise the reference SegPipeUNet methods that sit on the hot path (pipeline.py)."""
    for name in ("dask", "xarray"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "dask":
                m.config = types.SimpleNamespace(set=lambda **kw: None)
            sys.modules[name] = m
    import yaml
    from pipeline_train_predict.pipeline import SegPipeUNet
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "config_baseline.yaml")))
    cfg["save_model_params"] = False
    pipe = SegPipeUNet(experiment_name="golden", **cfg)
    pipe.device = torch.device("cpu")
    pipe.model.load_state_dict(sd)
    batch = {"data": x.double()}            # non-preload zarr path hands float64
    soft = pipe.predict_batch(batch, return_softmax=True)
    raw = pipe.predict_batch(batch, return_softmax=False)
    assert rel(raw, logits_eval) < 1e-6
    crit = pipe.get_criterion()
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=3,
                                              p=(0.85, 0.05, 0.05, 0.05))).long()
    loss = crit(raw, lab)
    o = orc.weighted_cross_entropy(raw, lab)
    assert abs(float(o) - float(loss)) < 1e-6 * abs(float(loss))
    raw_labels = torch.tensor([[-100, -70, -50, -30, -10, 0, 1, 2]], dtype=torch.int64)
    mapped = pipe.set_label_ignore_val(raw_labels.clone())
    assert torch.equal(mapped, orc.set_label_ignore_val(raw_labels))
    all_ign = crit(raw[:1, :, :4, :4], torch.full((1, 4, 4), -100, dtype=torch.int64))
    np.savez_compressed(
        os.path.join(OUT, "pipeline.npz"),
        softmax_ch12=soft[:, 1:3].numpy().astype(np.float32),
        ce_weight=crit.weight.numpy(), ce_loss=np.float64(loss),
        raw_labels=raw_labels.numpy(), mapped_labels=mapped.numpy(),
        all_ignored_is_nan=np.bool_(bool(torch.isnan(all_ign))),
    )


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    run_case("narrow8_64", start_filts=8, hw=64, keep_full_grads=True)
    sd, x, lab, logits_eval = run_case("full64_256", start_filts=64, hw=256, keep_full_grads=False)
    run_pipeline_case(sd, x, logits_eval)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
