#!/usr/bin/env python3
"""Development check of precision 'h3p' (pre-split fp16 plane pairs) on the GPU: golden parity of the whole net (eval
logits / argmax, train-mode logits, loss, every gradient vs the reference fixture) and a timing next to 'f32h3'."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth
import re
PRE_BN_BIAS = re.compile(r"(down_convs\.\d+\.main\.[03]|up_convs\.\d+\.conv[12])\.bias")


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def l2rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def golden(prec, scale=None):
    fix = np.load(os.path.join(ROOT, "tests", "golden", "full64_256.npz"))
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, 256, 256, seed=1)).cuda()
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=2)).cuda()
    m = pkg.UNet_Baseline(3, 4, precision=prec)
    m.load_state_dict(synth.synth_state_dict(seed=0))
    m.cuda().eval()
    if scale is not None:
        m.engine.loss_scale = scale
    with torch.no_grad():
        out = m(x)
    ref = torch.from_numpy(fix["logits_eval"])
    print(f"[{prec}] eval: rel {rel(out, ref):.3e}  flips {int((out.argmax(1).cpu() != ref.argmax(1)).sum())}")
    m.train()
    crit = pkg.WeightedCrossEntropy([10.0, 300.0, 250.0]).cuda()
    logits = m(x)
    loss = crit(logits, lab.long())
    loss.backward()
    torch.cuda.synchronize()
    ref_t = torch.from_numpy(fix["logits_train"])
    print(f"[{prec}] train: logits rel {rel(logits.detach(), ref_t):.3e} flips {int((logits.argmax(1).cpu() != ref_t.argmax(1)).sum())} "
          f"loss {float(loss):.8f} vs {float(fix['losses'][0]):.8f}  loss_scale {m.engine.loss_scale}")
    worst, rows = 0.0, []
    for k, p in m.named_parameters():
        if PRE_BN_BIAS.fullmatch(k):
            continue
        g = p.grad.detach().cpu()
        gn, noise = float(fix["gnorm/" + k]), float(fix["gnoise/" + k])
        nr = abs(float(g.double().norm()) - gn) / gn
        r = l2rel(g, fix["grad/" + k]) if "grad/" + k in fix.files else float("nan")
        rows.append((k, nr, r, noise))
        if r == r:
            worst = max(worst, r / max(noise, 1e-4))
    rows.sort(key=lambda t: -(t[2] if t[2] == t[2] else t[1]))
    for k, nr, r, noise in rows[:8]:
        print(f"    {k:34s} norm-rel {nr:.2e}  L2-rel {r:.2e}  (reference fp32-vs-fp64 noise {noise:.2e})")
    print(f"[{prec}] worst gradient L2-rel / noise: {worst:.1f}; finite: {all(torch.isfinite(p.grad).all() for p in m.parameters())}")


def timing(prec, B=32, steps=6):
    m = pkg.UNet_Baseline(3, 4, precision=prec)
    m.load_state_dict(synth.synth_state_dict(seed=0))
    m.cuda().train()
    eng = m.engine
    x = torch.from_numpy(synth.synth_echogram_batch(B, 4, 256, 256, seed=100)).cuda()
    lab = torch.from_numpy(synth.synth_labels(B, 256, 256, seed=200)).cuda()
    cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
    for _ in range(2):
        loss = eng.train_step(x, lab, cw, lr=0.005, momentum=0.95)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = eng.train_step(x, lab, cw, lr=0.005, momentum=0.95)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m.eval()
    with torch.no_grad():
        for _ in range(2):
            m.predict_softmax(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.predict_softmax(x)
        torch.cuda.synchronize()
    di = (time.perf_counter() - t0) / steps
    print(f"[{prec}] B={B}: train {1e3 * dt:.2f} ms/step = {B / dt:.0f} patches/s (loss {float(loss):.5f}, skipped "
          f"{eng.skipped_steps()}); infer {1e3 * di:.2f} ms = {B / di:.0f} patches/s")


if __name__ == "__main__":
    what = sys.argv[1:] or ["golden", "timing"]
    if "golden" in what:
        golden("f32h3")
        golden("h3p")
    if "timing" in what:
        timing("f32h3")
        timing("h3p")
