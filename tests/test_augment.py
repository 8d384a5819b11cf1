"""On-GPU augmentation + data transform (BASELINE configs[4]).

CPU: the Philox restatement against the Random123 known-answer vectors; the oracle's distributions
against what the reference's add_noise / flip_x_axis specify (and against the reference functions
themselves when /root/reference is present).  GPU: kernel == oracle on the same seed."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import augment_oracle as aug  # noqa: E402


def test_philox_known_answer_vectors():
    """Random123 kat_vectors for philox4x32-10."""
    def run(c, k):
        return [int(v[0]) for v in aug.philox4x32_10([c[0]], [c[1]], [c[2]], [c[3]], k[0], k[1])]
    assert run((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert run((f, f, f, f), (f, f)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_oracle_distributions_match_the_reference_specification():
    rng = np.random.default_rng(0)
    B, C, H, W = 64, 4, 32, 48
    data = np.ones((B, C, H, W), dtype=np.float32)        # linear sv = 1 -> 0 dB; ratios visible in dB
    labels = rng.integers(0, 3, (B, H, W)).astype(np.int16)
    out, lab, noisy, flipped = aug.augment_db(data, labels, seed=1234)
    assert 0.3 < noisy.mean() < 0.7 and 0.3 < flipped.mean() < 0.7          # p = .5 per sample
    assert np.all(out[~noisy] == 0.0)                                        # untouched samples: 0 dB
    lin = 10.0 ** (out[noisy].astype(np.float64) / 10.0)                     # clamp at 0 dB hides factors > 1
    changed = np.abs(out[noisy]) > 0
    # factors > 1 clamp to 0 dB, so only the "reduced" half (U(0,1)) is visible: 2.5 % of the values
    assert abs(changed.mean() - 0.025) < 0.004
    red = lin[changed]
    assert 0.0 <= red.min() and red.max() <= 1.0 and abs(np.median(red) - 0.5) < 0.05     # U(0,1)
    for b in range(B):
        ref = labels[b][:, ::-1] if flipped[b] else labels[b]
        assert np.array_equal(lab[b], ref)
    # increased half: use small inputs so the x U(1,10) factors stay below the clamp
    data2 = np.full((B, C, H, W), 1e-3, dtype=np.float32)
    out2, _, noisy2, _ = aug.augment_db(data2, labels, seed=99, do_flip=False)
    ratio = 10.0 ** ((out2[noisy2].astype(np.float64) + 30.0) / 10.0)
    ch = np.abs(ratio - 1.0) > 1e-4
    assert abs(ch.mean() - 0.05) < 0.006                                     # 5 % of the values change
    inc = ratio[ch & (ratio > 1.0)]
    assert abs(len(inc) / ch.sum() - 0.5) < 0.03 and inc.max() <= 10.0 + 1e-3 and abs(inc.mean() - 5.5) < 0.2


@pytest.mark.skipif(not os.path.isdir("/root/reference/crimac_unet"), reason="reference not present")
def test_reference_add_noise_has_the_same_distribution():
    sys.path.insert(0, "/root/reference/crimac_unet")
    from batch.data_augmentation.add_noise import add_noise
    np.random.seed(3)
    n_noisy, frac, inc = 0, [], []
    for _ in range(200):
        d = np.full((4, 32, 32), 1e-3)
        out, _, _ = add_noise(d.copy(), None, None)
        r = out / 1e-3
        ch = np.abs(r - 1) > 1e-9
        if ch.any():
            n_noisy += 1
            frac.append(ch.mean())
            inc.append((r[ch] > 1).mean())
    assert 70 < n_noisy < 130 and abs(np.mean(frac) - 0.05) < 0.005 and abs(np.mean(inc) - 0.5) < 0.03


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32x6", "bf16", "h3p"])
def test_gpu_augment_equals_oracle(prec):
    import crimac_classifiers_unet_amd as pkg
    rng = np.random.default_rng(5)
    B, C, H, W = 6, 4, 32, 48
    data = np.power(10.0, rng.uniform(-7.5, 0.0, (B, C, H, W))).astype(np.float32)
    data[0, 0, 3, 5] = np.nan
    data[1, 2, 7, 7] = np.inf
    labels = rng.integers(0, 3, (B, H, W)).astype(np.int64)
    ref, ref_lab, noisy, flipped = aug.augment_db(data, labels, seed=0xC0FFEE1234)
    assert noisy.any() and (~noisy).any() and flipped.any() and (~flipped).any()
    m = pkg.UNet_Baseline(3, 4, precision=prec).cuda()
    x, lab = m.engine.augment_batch(torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda(), 0xC0FFEE1234)
    if prec == "h3p":          # fp16 plane pairs: every 8-channel group holds [8 hi][8 lo], value = hi + lo
        h = x.cpu().view(torch.float16).view(-1, 2, 2, 8).float()
        got = (h[:, :, 0] + h[:, :, 1]).reshape(B, H, W, 16).numpy()
    else:
        got = x.float().cpu().numpy().reshape(B, H, W, 16)
    assert np.abs(got[..., 4:]).max() == 0
    tol = 0.3 if prec == "bf16" else (4e-5 if prec == "h3p" else 2e-5)
    assert np.abs(got[..., :4].transpose(0, 3, 1, 2) - ref).max() < tol
    assert np.array_equal(lab.cpu().numpy(), ref_lab)
    # a training step on raw crops runs and lowers the loss on repetition
    from crimac_classifiers_unet_amd import synth
    m.load_state_dict(synth.synth_state_dict(seed=0))
    cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
    d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    l1 = m.engine.train_step_augmented(d, l, cw, 0.005, 0.95, seed=7)
    l2 = m.engine.train_step_augmented(d, l, cw, 0.005, 0.95, seed=7)
    assert bool(torch.isfinite(l1)) and float(l2) < float(l1)


@pytest.mark.gpu
def test_gpu_augment_with_label_transform_equals_oracle_chain():
    """RAW annotation ids in: augmentation (noise + flip) -> refine_label_boundary + convert_label_indexing on the
    AUGMENTED linear crop -> NaN rule, the order of the reference's Dataset (batch/dataset.py:89-103)."""
    import crimac_classifiers_unet_amd as pkg
    from oracle import labels_oracle as lab_orc
    rng = np.random.default_rng(8)
    B, C, H, W = 6, 4, 64, 80
    data = np.power(10.0, rng.uniform(-9.0, -2.0, (B, C, H, W))).astype(np.float32)
    labels = np.zeros((B, H, W), dtype=np.int16)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(B):
        for k in range(4):
            cy, cx = rng.integers(0, H), rng.integers(0, W)
            blob = ((yy - cy) / rng.integers(4, 14)) ** 2 + ((xx - cx) / rng.integers(4, 20)) ** 2 <= 1
            labels[b][blob] = [27, 1, 12, 27][k]
            strong = blob & (rng.random((H, W)) < 0.8)
            data[b, 3][strong] = np.power(10.0, rng.uniform(-6.9, -4.1, int(strong.sum()))).astype(np.float32)
    labels[1, :, :9] = -100
    labels[2, 50:] = -100
    data[0, 0, 3:6, 5:40] = np.nan
    data[3, 3, 20:24, 10:30] = np.inf
    seed = 0xABCDEF0123
    _, raw_lab, noisy, flipped, lin = aug.augment_db(data, labels, seed=seed, return_linear=True)
    assert noisy.any() and flipped.any() and (~flipped).any()
    expect = lab_orc.train_label_transform(lin, raw_lab, 3)
    # the refine step must matter in this test, and depend on the noise (thresholds see augmented values)
    plain = np.stack([lab_orc.convert_label_indexing(raw_lab[b]) for b in range(B)])
    assert (expect != plain).sum() > 100
    m = pkg.UNet_Baseline(3, 4, precision="bf16").cuda()
    _, lab = m.engine.augment_batch(torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda(), seed,
                                    refine_labels=(3, 1e-7, 1e-4))
    assert np.array_equal(lab.cpu().numpy(), expect)
    assert set(np.unique(expect)) <= {0, 1, 2, -100}
