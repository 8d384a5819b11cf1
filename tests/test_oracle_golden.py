"""CPU: the oracle (oracle/unet_oracle.py) against golden vectors captured from the imported
reference (tools/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare
the HIP path with the oracle and with the same fixtures."""
import os
import re

import numpy as np
import pytest
import torch

from crimac_classifiers_unet_amd import synth
from oracle import unet_oracle as orc

PRE_BN_BIAS = re.compile(r"down_convs\.\d+\.main\.[03]\.bias|up_convs\.\d+\.conv[12]\.bias")


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def _case(golden_dir, tag):
    fix = np.load(os.path.join(golden_dir, tag + ".npz"))
    sf, hw = int(fix["start_filts"]), int(fix["hw"])
    sd = synth.synth_state_dict(start_filts=sf, seed=0)
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, hw, hw, seed=1))
    lab = torch.from_numpy(synth.synth_labels(2, hw, hw, seed=2))
    return fix, sd, x, lab


@pytest.mark.parametrize("tag", ["narrow8_64", "full64_256"])
def test_eval_logits_match_reference(golden_dir, tag):
    fix, sd, x, _ = _case(golden_dir, tag)
    torch.set_num_threads(8)
    out = orc.predict(sd, x)
    assert _rel(out, fix["logits_eval"]) < 5e-6
    assert np.array_equal(out.argmax(1).numpy(), fix["logits_eval"].argmax(1))


def test_train_step_matches_reference_narrow(golden_dir):
    fix, sd, x, lab = _case(golden_dir, "narrow8_64")
    loss, logits, grads, stats = orc.loss_and_grads(sd, x, lab)
    assert _rel(logits, fix["logits_train"]) < 5e-6
    assert abs(float(loss) - fix["losses"][0]) < 1e-5
    for k, g in grads.items():
        if PRE_BN_BIAS.fullmatch(k):
            continue
        ref = torch.from_numpy(fix["grad/" + k])
        assert float((g - ref).norm() / ref.norm()) < 1e-4, k
    for k, v in stats.items():
        assert _rel(v.float(), fix["stat1/" + k]) < 1e-5, k
    state, losses = orc.train_steps(sd, [(x, lab)] * 3, lr=0.005, momentum=0.95)
    assert np.allclose(losses, fix["losses"], rtol=2e-4)
    for k, v in state.items():
        assert abs(float(v.double().norm()) - float(fix["final_norm/" + k])) <= 2e-4 * float(fix["final_norm/" + k]) + 1e-12, k


def test_train_step_matches_reference_full(golden_dir):
    fix, sd, x, lab = _case(golden_dir, "full64_256")
    torch.set_num_threads(8)
    loss, logits, grads, _ = orc.loss_and_grads(sd, x, lab)
    assert _rel(logits, fix["logits_train"]) < 1e-5
    assert abs(float(loss) - fix["losses"][0]) < 1e-5
    # fp32 gradient noise floor of the reference itself (vs its fp64 run) is stored per key
    for k, g in grads.items():
        if PRE_BN_BIAS.fullmatch(k):
            continue
        gn, noise = float(fix["gnorm/" + k]), float(fix["gnoise/" + k])
        assert abs(float(g.double().norm()) - gn) <= max(4 * noise, 1e-3) * gn, k
        if "grad/" + k in fix.files:
            ref = torch.from_numpy(fix["grad/" + k])
            assert float((g - ref).norm() / ref.norm()) < max(4 * noise, 1e-3), k


def test_pipeline_pieces(golden_dir):
    fix = np.load(os.path.join(golden_dir, "pipeline.npz"))
    sd = synth.synth_state_dict(start_filts=64, seed=0)
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, 256, 256, seed=1))
    torch.set_num_threads(8)
    soft = orc.predict(sd, x.double(), return_softmax=True)
    assert np.abs(soft[:, 1:3].numpy() - fix["softmax_ch12"]).max() < 2e-6
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=3, p=(0.85, 0.05, 0.05, 0.05)))
    loss = orc.weighted_cross_entropy(orc.predict(sd, x), lab)
    assert abs(float(loss) - float(fix["ce_loss"])) < 1e-5 * abs(float(fix["ce_loss"]))
    assert tuple(fix["ce_weight"]) == orc.CE_CLASS_WEIGHTS
    mapped = orc.set_label_ignore_val(torch.from_numpy(fix["raw_labels"]))
    assert np.array_equal(mapped.numpy(), fix["mapped_labels"])
    all_ign = orc.weighted_cross_entropy(torch.zeros(1, 3, 4, 4), torch.full((1, 4, 4), -100))
    assert bool(torch.isnan(all_ign)) == bool(fix["all_ignored_is_nan"])


def test_state_shapes_match_reference_layout():
    shapes = synth.unet_state_shapes()
    assert len(shapes) == 136
    n_params = sum(int(np.prod(s)) for k, s in shapes.items()
                   if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")))
    assert n_params == 31044227      # SURVEY.md §5 / BASELINE.md
    assert shapes["up_convs.0.upconv.weight"] == (1024, 512, 2, 2)
    assert shapes["up_convs.0.conv1.weight"] == (512, 1024, 3, 3)
    assert shapes["conv_final.weight"] == (3, 64, 1, 1)


def test_storage_rounding_oracle_is_the_fp32_oracle_up_to_the_16bit_envelope():
    """oracle/unet_lowp_oracle.py (bf16 / fp16 storage points of the HIP engine) stays within the 16-bit envelope
    of the fp32 oracle on a small case, rounds exactly where it says, and leaves fp32 weights untouched."""
    import torch
    from crimac_classifiers_unet_amd import synth
    from oracle import unet_lowp_oracle as lowp
    from oracle import unet_oracle as orc
    sd = synth.synth_state_dict(seed=3)
    x = torch.from_numpy(synth.synth_echogram_batch(1, 4, 32, 32, seed=5))
    lab = torch.from_numpy(synth.synth_labels(1, 32, 32, seed=6))
    ref_loss, ref_logits, ref_grads, _ = orc.loss_and_grads(sd, x, lab)
    for storage, tol in (("bf16", 8e-2), ("fp16", 1e-2)):
        loss, logits, grads, stats = lowp.loss_and_grads(sd, x, lab, storage=storage)
        assert float((logits - ref_logits).abs().max() / ref_logits.abs().max()) < tol
        assert abs(float(loss) - float(ref_loss)) < tol * abs(float(ref_loss))
        assert set(grads) == set(ref_grads) and all(torch.isfinite(g).all() for g in grads.values())
        g, r = grads["conv_final.weight"], ref_grads["conv_final.weight"]
        assert float((g - r).norm() / r.norm()) < 5 * tol
        assert len(stats) == 36
    v = torch.randn(1000)
    q = lowp._st(v.clone().requires_grad_(True), torch.bfloat16)
    assert torch.equal(q.detach(), v.to(torch.bfloat16).float())
    w = torch.randn(8, requires_grad=True)
    (lowp._wq(w, torch.bfloat16) * torch.arange(8.0)).sum().backward()
    assert torch.equal(w.grad, torch.arange(8.0))
