"""Training label transform (refine_label_boundary + convert_label_indexing + the NaN rule): the numpy oracle
against vectors produced by the reference itself (tools/make_golden_labels.py), and the HIP kernel against
the oracle."""
import os

import numpy as np
import pytest

from oracle import labels_oracle as orc


def _cases(golden_dir):
    fix = np.load(os.path.join(golden_dir, "labels.npz"))
    n = len({k.split("/")[0] for k in fix.files})
    for i in range(n):
        d03 = fix[f"c{i}/data03"]
        data = np.zeros((4,) + d03.shape[1:], dtype=np.float32)
        data[0], data[3] = d03[0], d03[1]
        yield i, data, fix[f"c{i}/labels"], fix[f"c{i}/after_label_transform"], fix[f"c{i}/final"]


def test_oracle_matches_reference_label_transform(golden_dir):
    seen_refined = 0
    for i, data, labels, after_lt, final in _cases(golden_dir):
        refined = orc.refine_label_boundary(data[3], labels)
        got = orc.convert_label_indexing(refined).astype(np.int16)
        assert np.array_equal(got, after_lt), f"case {i}: label transform differs at {np.argwhere(got != after_lt)[:5]}"
        full = orc.train_label_transform(data[None], labels[None], 3)[0]
        assert np.array_equal(full, final), f"case {i}: final labels differ"
        assert set(np.unique(full)) <= {0, 1, 2, -100}
        seen_refined += int(((labels > 0) & (refined == orc.LABEL_REFINE_BOUNDARY_VAL)).sum())
    assert seen_refined > 1000          # the fixtures do exercise the closing


def test_closing_equals_scipy():
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(3)
    for shape in [(1, 1), (3, 9), (7, 7), (20, 31), (64, 64)]:
        for p in (0.1, 0.5, 0.9):
            m = rng.random(shape) < p
            assert np.array_equal(orc.binary_closing_disk7(m), ndi.binary_closing(m, structure=orc.CLOSING))


@pytest.mark.gpu
def test_refine_labels_kernel_matches_oracle(golden_dir):
    import torch
    from crimac_classifiers_unet_amd.hip import call, ptr
    for i, data, labels, _, final in _cases(golden_dir):
        H, W = labels.shape
        d = torch.from_numpy(data[None]).cuda()
        for dt in (torch.int16, torch.int32, torch.int64):
            lab = torch.from_numpy(labels[None].astype(np.int64)).to(dt).cuda()
            out = torch.full((1, H, W), 7, dtype=torch.int16, device="cuda")
            call("crimac_refine_labels", ptr(lab), lab.element_size(), None, ptr(d), 3, 1e-7, 1e-4, 1, ptr(out),
                 1, 4, H, W)
            torch.cuda.synchronize()
            assert np.array_equal(out[0].cpu().numpy(), final), (i, dt)
    # a batch of different patches in one launch + the aux-mask input form (threshold / NaN bits precomputed)
    rng = np.random.default_rng(0)
    B, H, W = 5, 64, 80
    data = (10.0 ** rng.uniform(-9, -2, size=(B, 4, H, W))).astype(np.float32)
    labels = rng.choice(np.array([0, 0, 0, 27, 1, 12, -100], dtype=np.int16), size=(B, H, W))
    labels[1, :, :11] = -100
    labels[2] = -100
    data[3, 0, 5:9, 7:30] = np.nan
    expect = orc.train_label_transform(data, labels, 3)
    with np.errstate(invalid="ignore"):
        aux = ((data[:, 3] > np.float32(1e-7)) & (data[:, 3] < np.float32(1e-4))).astype(np.uint8) \
            | ((~np.isfinite(data[:, 0])).astype(np.uint8) << 1)
    lab_d, aux_d, data_d = torch.from_numpy(labels).cuda(), torch.from_numpy(aux).cuda(), torch.from_numpy(data).cuda()
    for use_aux in (False, True):
        out = torch.empty(B, H, W, dtype=torch.int16, device="cuda")
        call("crimac_refine_labels", ptr(lab_d), 2, ptr(aux_d) if use_aux else None, None if use_aux else ptr(data_d),
             3, 1e-7, 1e-4, 1, ptr(out), B, 4, H, W)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), expect), use_aux
    # mode 0: refine only (raw ids kept, -30 marks)
    out = torch.empty(B, H, W, dtype=torch.int16, device="cuda")
    call("crimac_refine_labels", ptr(lab_d), 2, None, ptr(data_d), 3, 1e-7, 1e-4, 0, ptr(out), B, 4, H, W)
    torch.cuda.synchronize()
    ref0 = np.stack([orc.refine_label_boundary(data[b, 3], labels[b]) for b in range(B)])
    assert np.array_equal(out.cpu().numpy(), ref0)
