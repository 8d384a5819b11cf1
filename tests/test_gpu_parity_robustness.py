"""Parity away from i.i.d. noise and default-scale weights (VERDICT r4 weak #1c, #1d, #1b):
  * whole-net h3p eval against the CPU oracle on STRUCTURED crops -- annotated schools, a seabed echo line, whole pings of
    NaN / Inf that go through remove_nan_inf -- with the host transform and with the on-GPU transform;
  * the same on a TRAINED state: 100 bf16 steps on school labels on the GPU, the resulting state_dict handed to the oracle;
  * one h3f TRAINING step at B = 32 (BASELINE configs[1] size) against the oracle: loss, BatchNorm running statistics,
    gradients;
  * the single-pass distribution of the golden gradient errors (20 passes) instead of a median of three."""
import os

import numpy as np
import pytest
import torch

import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth
from oracle import tiling_oracle as torc, unet_oracle as orc

pytestmark = pytest.mark.gpu

TIE_MARGIN = 2e-6          # as tests/test_gpu_h3p.py: relative to logits of O(1); scaled by max |logit| below
MAX_TIE_FRACTION = 1e-5


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def _l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def assert_masks_match_up_to_ties(out, ref, what):
    diff = out.argmax(1) != ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])[diff]
    scale = max(1.0, float(ref.abs().max()))
    n = int(diff.sum())
    print(f"{what}: argmax flips={n}/{ref[:, 0].numel()} oracle margins at flips={[f'{v:.1e}' for v in margin.tolist()[:8]]} "
          f"(max |logit| {scale:.2f})")
    assert bool((margin < TIE_MARGIN * scale).all()), (what, float(margin.max()))
    assert n <= max(MAX_TIE_FRACTION * ref[:, 0].numel(), 1), (what, n)


def structured_survey(seed=5):
    """Schools (annotated, strong in the last channel), a seabed echo line (a 12-sample band of 10^U(-1.5, 0) at an
    undulating seabed, weak reverberation below it), whole pings of NaN and of +Inf, scattered NaN / Inf samples."""
    n_pings, n_range = 4096, 1024
    r = synth.SyntheticSurveyReader(n_pings=n_pings, n_range=n_range, block=n_pings, schools=80, bad_frac=2e-4, seed=seed,
                                    seabed_index=700)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = np.arange(n_pings)
    sb = (700 + 60 * np.sin(x / 150.0) + 15 * np.sin(x / 23.0)).astype(np.int64)
    r.seabed = sb
    rows = np.arange(n_range)[None, :]
    band = (rows >= sb[:, None]) & (rows < sb[:, None] + 12)
    below = rows >= sb[:, None] + 12
    for c in range(4):
        r.sv[c][band] = np.power(10.0, rng.uniform(-1.5, 0.0, size=int(band.sum()))).astype(np.float32)
        r.sv[c][below] = np.power(10.0, rng.uniform(-9.0, -7.0, size=int(below.sum()))).astype(np.float32)
    r.sv[:, 500:503] = np.nan                  # dropped pings
    r.sv[:, 1800] = np.inf
    r.sv[0, 2500:2502] = np.nan                # channel 0 only: the label rule's channel
    r.labels[below] = 0
    return r


_HW = {}


def crops(reader, n, seed, size=256):
    """n raw crops [4, size, size] + raw annotation ids; NaN and Inf samples are KEPT (the memmap flavour's crop zeroes
    them, batch/dataset.py:279-281; here they are remove_nan_inf's business): four fixed centres over a dropped ping, the
    Inf ping, the channel-0 NaN pings and the seabed line, the rest random."""
    if id(reader) not in _HW:
        _HW[id(reader)] = (np.ascontiguousarray(reader.sv.transpose(0, 2, 1)), np.ascontiguousarray(reader.labels.T))
    sv_hw, lab_hw = _HW[id(reader)]
    ds = synth.RawCropDataset(reader, (size, size), n, seed=seed)
    cen = [(600, 520), (700, 1790), (300, 2490), (650, 3000)][:n] + [tuple(ds.centre(i)) for i in range(4, n)]
    raw = np.stack([torc.crop(sv_hw, c, (size, size), 0) for c in cen]).astype(np.float32)
    labels = np.stack([torc.crop(lab_hw, c, (size, size), -100) for c in cen]).astype(np.int16)
    return raw, labels


def host_transform(raw):
    """remove_nan_inf + db_with_limits as the reference's workers apply them (oracle restatement)."""
    return np.stack([torc.data_transform(x)[0] for x in raw]).astype(np.float32)


def test_structured_crops_h3p_eval_matches_oracle_host_and_gpu_transform():
    reader = structured_survey()
    raw, _ = crops(reader, 8, seed=31)
    assert np.isnan(raw).any() and np.isinf(raw).any()
    x = torch.from_numpy(host_transform(raw))
    assert bool(torch.isfinite(x).all()) and float(x.min()) == -75.0 and -1.0 < float(x.max()) <= 0.0
    sd = synth.synth_state_dict(seed=0)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ref = orc.predict(sd, x)
    m = pkg.UNet_Baseline(3, 4, precision="h3p")
    m.load_state_dict(sd)
    m.cuda().eval()
    with torch.no_grad():
        out = m(x.cuda()).cpu()
        # the on-GPU data transform (crimac_augment_db_nhwc without augmentation) straight into the first convolution
        xg, _ = m.infer_engine.augment_batch(torch.from_numpy(raw).cuda(), None, 0, do_noise=False, do_flip=False)
        out_g = m.infer_engine.forward_nhwc(xg, 8, 256, 256, training=False).cpu()
    print(f"structured crops, h3p eval: rel {_rel(out, ref):.3e} (host transform) {_rel(out_g, ref):.3e} (GPU transform)")
    assert _rel(out, ref) < 1e-5
    assert_masks_match_up_to_ties(out, ref, "structured, host transform")
    # the GPU's log10 may differ from numpy's in the last bit of a dB value: logits within 1e-4, masks up to ties
    assert _rel(out_g, ref) < 1e-4
    d = out_g.argmax(1) != ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    assert int(d.sum()) <= 1e-4 * d.numel() and bool(((top2[:, 0] - top2[:, 1])[d] < 1e-4).all())


def test_trained_state_h3p_eval_matches_oracle():
    """100 bf16 training steps on raw school crops (augmentation + label refinement + dB on the GPU), then the TRAINED
    state_dict -- weights and BatchNorm statistics far from their initial scale -- goes to the CPU oracle."""
    reader = structured_survey(seed=9)
    m = pkg.UNet_Baseline(3, 4, precision="bf16", infer_precision="h3p")
    m.load_state_dict(synth.synth_state_dict(seed=0))
    m.cuda().train()
    eng = m.engine
    cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    losses = []
    for step in range(100):
        raw, lab = crops(reader, 8, seed=100 + step % 10)
        loss = eng.train_step_augmented(torch.from_numpy(raw).cuda(), torch.from_numpy(lab).cuda(), cw, 0.005, 0.95,
                                        seed=step, refine_labels=(3, 1e-7, 1e-4))
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-10:])
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    moved = _l2(sd["down_convs.2.main.0.weight"], sd0["down_convs.2.main.0.weight"].cpu())
    rv = sd["up_convs.3.bn2.running_var"]
    print(f"trained: loss {np.mean(losses[:5]):.3f} -> {np.mean(losses[-10:]):.3f}, a weight moved by {moved:.2e} (L2-rel), "
          f"running_var of the last block {float(rv.min()):.2e} .. {float(rv.max()):.2e}")
    assert moved > 1e-3
    raw, _ = crops(reader, 8, seed=77)
    x = torch.from_numpy(host_transform(raw))
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ref = orc.predict(sd, x)
    m.eval()
    with torch.no_grad():
        out = m(x.cuda()).cpu()          # the model's eval engine: h3p on the parameters the bf16 engine just trained
    print(f"trained state, h3p eval: rel {_rel(out, ref):.3e}, max |logit| {float(ref.abs().max()):.2f}, "
          f"classes predicted {np.bincount(ref.argmax(1).reshape(-1).numpy(), minlength=3).tolist()}")
    assert _rel(out, ref) < 2e-5
    assert_masks_match_up_to_ties(out, ref, "trained state")
    assert len(set(ref.argmax(1).reshape(-1).tolist())) >= 2          # (the trained net does predict schools)


def test_h3f_training_step_batch32_matches_oracle():
    """One h3f training step on 32 x 4 x 256 x 256 (the benchmark's shape and dispatch) against the oracle's step."""
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    sd = synth.synth_state_dict(seed=0)
    x = torch.from_numpy(synth.synth_echogram_batch(32, 4, 256, 256, seed=1))
    lab = torch.from_numpy(synth.synth_labels(32, 256, 256, seed=2))
    ref_loss, ref_logits, ref_grads, ref_stats = orc.loss_and_grads(sd, x, lab)
    m = pkg.UNet_Baseline(3, 4, precision="h3f")
    m.load_state_dict(sd)
    m.cuda().train()
    crit = pkg.WeightedCrossEntropy([10.0, 300.0, 250.0]).cuda()
    logits = m(x.cuda())
    loss = crit(logits, lab.long().cuda())
    loss.backward()
    assert _rel(logits.detach(), ref_logits) < 2e-5
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    for k, v in ref_stats.items():
        if "running" in k:
            assert _rel(m.state_dict()[k].float(), v) < 1e-4, k
    g = {k: p.grad for k, p in m.named_parameters()}
    worst = {}
    for k in ("conv_final.weight", "down_convs.0.main.0.weight", "down_convs.4.main.3.weight", "up_convs.0.upconv.weight",
              "up_convs.3.conv2.weight", "down_convs.2.main.1.weight"):
        worst[k] = _l2(g[k], ref_grads[k])
    print("h3f B=32 training step vs oracle: loss rel "
          f"{abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)):.2e}; gradient L2-rel " +
          ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    # the reference's own fp32-vs-fp64 gradient noise is 3-4e-3 L2 per tensor on the golden crops (tests/golden gnoise/*);
    # the golden test holds h3f to max(6 x noise, 3e-3) per tensor at B = 2
    for k, v in worst.items():
        assert v < 2.5e-2, (k, v)
    assert m.engine.skipped_steps() == 0


@pytest.mark.parametrize("precision", ["h3p", "h3f"])
def test_golden_gradient_error_distribution_single_passes(precision, golden_dir):
    """VERDICT r4 weak #1b: not a median of three -- the DISTRIBUTION of the single-pass error of every gradient tensor of
    the golden training step over 20 passes (each a fresh forward + backward of the same model on the same crops; passes
    differ because BatchNorm sums are added up by atomics in arrival order, which now and then moves a ReLU mask or a pool
    position).  Asserted: the 90th percentile of every tensor within the tolerance max(6 x the reference's own fp32-vs-fp64
    noise of that tensor, 3e-3), and no single pass beyond 1.5 x it.  The table is printed (pytest -s) and committed as
    profiles/r05_golden_gradient_error_distribution.txt."""
    import re
    pre_bn_bias = re.compile(r"(down_convs\.\d+\.main\.[03]|up_convs\.\d+\.conv[12])\.bias")
    fix = np.load(os.path.join(golden_dir, "full64_256.npz"))
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, 256, 256, seed=1)).cuda()
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=2)).cuda()
    m = pkg.UNet_Baseline(3, 4, precision=precision)
    m.load_state_dict(synth.synth_state_dict(seed=0))
    m.cuda().train()
    crit = pkg.WeightedCrossEntropy([10.0, 300.0, 250.0]).cuda()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    errs = {}
    N = 20
    for _ in range(N):
        m.load_state_dict(sd0)               # (the training forward moves the running statistics; gradients do not care)
        for p_ in m.parameters():
            p_.grad = None
        loss = crit(m(x), lab.long())
        loss.backward()
        for k, p_ in m.named_parameters():
            if pre_bn_bias.fullmatch(k) or "grad/" + k not in fix.files:
                continue
            errs.setdefault(k, []).append(_l2(p_.grad.detach().cpu(), fix["grad/" + k]))
    rows, worst_p90, worst_max = [], 0.0, 0.0
    for k, v in errs.items():
        tol = max(6 * float(fix["gnoise/" + k]), 3e-3)
        q = np.quantile(np.array(v) / tol, [0.5, 0.9, 1.0])
        rows.append((q[1], k, tol, q))
        worst_p90, worst_max = max(worst_p90, q[1]), max(worst_max, q[2])
    rows.sort(reverse=True)
    print(f"\n{precision}: single-pass gradient L2-rel error / tolerance over {N} passes, {len(rows)} tensors "
          f"(worst p90 {worst_p90:.2f}, worst single pass {worst_max:.2f})")
    print(f"{'tensor':44s} {'tol':>9s} {'p50':>6s} {'p90':>6s} {'max':>6s}")
    for _, k, tol, q in rows[:12]:
        print(f"{k:44s} {tol:9.2e} {q[0]:6.2f} {q[1]:6.2f} {q[2]:6.2f}")
    assert worst_p90 <= 1.0, rows[0]
    assert worst_max <= 1.5, rows[0]
