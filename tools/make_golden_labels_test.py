#!/usr/bin/env python3
"""Golden vectors for the TEST-time label transform, produced by the REFERENCE's own define_label_transform_test
(batch/transforms.py:81-99: convert_label_indexing_unused_species, refine_label_boundary, mask_label_seabed,
mask_label_overlap) followed by its data transform's label rule (remove_nan_inf), run on crops of a synthetic survey
held by an in-memory reader with the real readers' get_seabed_mask semantics (tools/fake_reader.py).
Build container only (imports /root/reference)."""
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

from batch.transforms import define_label_transform_test, define_data_transform  # noqa: E402  (reference)
from oracle import tiling_oracle as torc  # noqa: E402  (crop only: inputs of the transform, pinned by tiling.npz)
from tools.fake_reader import FakeEchogram, FakeZarrReader, holey_seabed_mask  # noqa: E402


def survey(rng, n_pings, n_range):
    """linear sv [4, pings, range], raw labels [pings, range] with school blobs around the refine thresholds, seabed."""
    sv = (10.0 ** rng.uniform(-9.0, -2.0, size=(4, n_pings, n_range))).astype(np.float32)
    labels = np.zeros((n_pings, n_range), dtype=np.int64)
    xx, yy = np.mgrid[0:n_pings, 0:n_range]
    for k in range(40):
        cx, cy = rng.integers(0, n_pings), rng.integers(0, n_range)
        rx, ry = rng.integers(4, 30), rng.integers(3, 22)
        blob = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
        labels[blob] = [27, 1, 27, 12, 5000, -1][k % 6]
        strong = blob & (rng.random(labels.shape) < 0.8)
        sv[3][strong] = (10.0 ** rng.uniform(-6.9, -4.1, size=int(strong.sum()))).astype(np.float32)
    sv[0][rng.random(labels.shape) < 1e-3] = np.nan
    sv[3][rng.random(labels.shape) < 5e-4] = np.inf
    x = np.arange(n_pings)
    seabed = np.clip((0.7 * n_range + 0.15 * n_range * np.sin(x / 41.0)).astype(np.int64), 30, n_range - 5)
    return sv, labels, seabed


def main():
    rng = np.random.default_rng(11)
    freqs = [18, 38, 120, 200]
    n_pings, n_range = 520, 330
    sv, labels, seabed = survey(rng, n_pings, n_range)
    dt = define_data_transform()
    out = {"sv03": sv[[0, 3]], "labels": labels.astype(np.int16), "seabed": seabed.astype(np.int32)}
    mask = holey_seabed_mask(seabed, n_range)
    out["holey_mask"] = mask
    readers = {"zarr": FakeZarrReader(sv, labels, seabed), "zarrmask": FakeZarrReader(sv, labels, seabed, mask=mask),
               "memm": FakeEchogram(np.ascontiguousarray(sv.transpose(0, 2, 1)), np.ascontiguousarray(labels.T), seabed)}
    cases = []          # (flavour, patch size, overlap, centre (range, ping))
    for flavour in readers:
        for (size, overlap) in ((64, 0), (96, 20), (128, 12)):
            for centre in ((size // 2 - 1, size // 2 - 1), (n_range - 20, 115), (int(seabed[300]) - 5, 300),
                           (40, n_pings - 10), (int(seabed[120]) + 8, 125), (5, 3), (n_range + 10, 200)):
                cases.append((flavour, size, overlap, centre))
    meta = []
    for i, (flavour, size, overlap, centre) in enumerate(cases):
        reader = readers[flavour]
        lt = define_label_transform_test(freqs, label_masks="all", patch_overlap=overlap)
        c = {"local": centre, "global": centre}
        # crops as the reference Dataset builds them (range-major patches, -100 / 0 outside the data)
        lab = torc.crop(np.ascontiguousarray(labels.T), centre, (size, size), -100).astype(np.int64)
        data = torc.crop(np.ascontiguousarray(sv.transpose(0, 2, 1)), centre, (size, size), 0).astype(np.float32)
        d1, l1, _, _ = lt(data.copy(), lab.copy(), np.array(centre), reader)
        _, l2, _, _ = dt(d1.copy(), np.asarray(l1).copy(), reader, freqs)
        out[f"c{i}/final"] = np.asarray(l2).astype(np.int16)
        meta.append((["zarr", "zarrmask", "memm"].index(flavour), size, overlap, centre[0], centre[1]))
        vals, cnt = np.unique(out[f"c{i}/final"], return_counts=True)
        print(i, flavour, size, overlap, centre, dict(zip(vals.tolist(), cnt.tolist())))
    out["cases"] = np.array(meta, dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "labels_test.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
