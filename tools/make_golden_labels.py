#!/usr/bin/env python3
"""Golden vectors for the training label transform, produced by the REFERENCE's own
define_label_transform_train (batch/transforms.py:71-78) followed by its define_data_transform label rule
(remove_nan_inf), run on synthetic crops.  Build container only (imports /root/reference)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/crimac_unet")
for name in ("dask", "xarray", "numcodecs", "tqdm"):
    try:
        __import__(name)
    except Exception:
        m = types.ModuleType(name)
        if name == "dask":
            m.config = types.SimpleNamespace(set=lambda **kw: None)
        sys.modules[name] = m

from batch.transforms import define_label_transform_train, define_data_transform  # noqa: E402  (reference)


def synth_case(rng, H, W, boundary, nan_frac=0.0):
    """Linear sv crop [4,H,W] float32 with school-like blobs around the refine thresholds, raw labels int16."""
    data = (10.0 ** rng.uniform(-9.0, -2.0, size=(4, H, W))).astype(np.float32)
    labels = np.zeros((H, W), dtype=np.int16)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(rng.integers(3, 9)):
        cy, cx = rng.integers(0, H), rng.integers(0, W)
        ry, rx = rng.integers(3, max(4, H // 5)), rng.integers(3, max(4, W // 4))
        blob = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        labels[blob] = [27, 1, 27, 12, 5000][k % 5]
        # inside a school most pixels are strong (between the thresholds), with weak holes and a weak rim
        strong = blob & (rng.random((H, W)) < 0.8)
        data[3][strong] = (10.0 ** rng.uniform(-6.9, -4.1, size=int(strong.sum()))).astype(np.float32)
    t, b_, l, r = boundary
    if t:
        labels[:t] = -100
    if b_:
        labels[H - b_:] = -100
    if l:
        labels[:, :l] = -100
    if r:
        labels[:, W - r:] = -100
    if nan_frac > 0:
        bad = rng.random((H, W)) < nan_frac
        data[0][bad] = np.nan
        data[3][rng.random((H, W)) < nan_frac] = np.inf
    return data, labels


def main():
    rng = np.random.default_rng(7)
    freqs = [18, 38, 120, 200]
    lt = define_label_transform_train(freqs)
    dt = define_data_transform()
    cases = [(96, 96, (0, 0, 0, 0), 0.0), (96, 80, (10, 0, 0, 25), 0.0), (64, 96, (0, 30, 17, 0), 0.01),
             (48, 48, (48, 0, 0, 0), 0.0), (256, 256, (0, 40, 0, 0), 0.002), (40, 72, (3, 3, 3, 3), 0.0)]
    out = {}
    for i, (H, W, bnd, nf) in enumerate(cases):
        data, labels = synth_case(rng, H, W, bnd, nf)
        echogram = types.SimpleNamespace(name="synthetic")
        d1, l1, _, _ = lt(data.copy(), labels.copy(), (H // 2, W // 2), echogram)
        _, l2, _, _ = dt(d1.copy(), l1.copy(), echogram, freqs)
        out[f"c{i}/data03"] = data[[0, 3]]           # the label rules only look at channels 0 (NaN) and 3 (threshold)
        out[f"c{i}/labels"] = labels
        out[f"c{i}/after_label_transform"] = np.asarray(l1).astype(np.int16)
        out[f"c{i}/final"] = np.asarray(l2).astype(np.int16)
        print(i, (H, W), "refined to ignore:", int(((labels > 0) & (np.asarray(l1) == -100)).sum()),
              "labels>0:", int((labels > 0).sum()))
    path = os.path.join(ROOT, "tests", "golden", "labels.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
