#!/usr/bin/env python3
"""Golden vectors for eval_mode 'region' / 'trace': the REFERENCE's define_label_transform_test(label_masks=...)
(batch/transforms.py:81-99 with get_extended_label_mask_for_crop, batch/label_transforms/extend_label_masks.py:35-98)
followed by its data transform's label rule, on crops of the synthetic survey of tools/make_golden_labels_test.py (same
seed: the survey arrays live in tests/golden/labels_test.npz and are not stored twice) held by in-memory readers that offer
``get_object_bounding_boxes``.  Build container only (imports /root/reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import make_golden_labels_test as base  # noqa: E402  (sets up the reference import + stubs)
from batch.transforms import define_label_transform_test, define_data_transform  # noqa: E402  (reference)
from oracle import labels_oracle as lo, tiling_oracle as torc  # noqa: E402
from tools.fake_reader import FakeEchogram, FakeZarrReader  # noqa: E402


def main():
    rng = np.random.default_rng(11)
    freqs = [18, 38, 120, 200]
    n_pings, n_range = 520, 330
    sv, labels, seabed = base.survey(rng, n_pings, n_range)
    stored = np.load(os.path.join(ROOT, "tests", "golden", "labels_test.npz"))
    assert np.array_equal(stored["labels"], labels.astype(np.int16)) and np.array_equal(stored["sv03"], sv[[0, 3]], equal_nan=True)
    brng = np.random.default_rng(23)
    boxes = []
    for _ in range(30):                  # (y0, y1, x0, x1), data_reader.py:112: some tiny, some large, some at the rim
        y0, x0 = int(brng.integers(-5, n_range - 5)), int(brng.integers(-5, n_pings - 5))
        boxes.append((y0, y0 + int(brng.integers(1, 60)), x0, x0 + int(brng.integers(1, 90))))
    boxes = np.array(boxes, dtype=np.int64)
    dt = define_data_transform()
    readers = {"zarr": FakeZarrReader(sv, labels, seabed, boxes=boxes),
               "memm": FakeEchogram(np.ascontiguousarray(sv.transpose(0, 2, 1)), np.ascontiguousarray(labels.T), seabed,
                                    boxes=boxes)}
    out, meta, i = {"boxes": boxes.astype(np.int32)}, [], 0
    for flavour, reader in readers.items():
        for mask_type in ("region", "trace"):
            for (size, overlap, extend) in ((64, 0, 20), (128, 12, 7)):
                for centre in ((size // 2 - 1, size // 2 - 1), (n_range - 20, 115), (int(seabed[300]) - 5, 300),
                               (40, n_pings - 10), (150, 260), (5, 3)):
                    lt = define_label_transform_test(freqs, label_masks=mask_type, extend_size=extend, patch_overlap=overlap)
                    lab = torc.crop(np.ascontiguousarray(labels.T), centre, (size, size), -100).astype(np.int64)
                    data = torc.crop(np.ascontiguousarray(sv.transpose(0, 2, 1)), centre, (size, size), 0).astype(np.float32)
                    d1, l1, _, _ = lt(data.copy(), lab.copy(), np.array(centre), reader)
                    _, l2, _, _ = dt(d1.copy(), np.asarray(l1).copy(), reader, freqs)
                    ref = np.asarray(l2).astype(np.int16)
                    # the oracle against the reference, here and now
                    bx = lo.extend_boxes(boxes, mask_type, extend, reader.shape[0])
                    got = lo.test_label_transform(data, lab, centre, 3, seabed, n_range, overlap,
                                                  seabed_rule="zarr" if flavour == "zarr" else "memm", boxes_extended=bx)
                    assert np.array_equal(got, ref), (flavour, mask_type, size, centre)
                    out[f"c{i}/final"] = ref
                    meta.append((["zarr", "memm"].index(flavour), ["region", "trace"].index(mask_type), size, overlap, extend,
                                 centre[0], centre[1]))
                    vals, cnt = np.unique(ref, return_counts=True)
                    print(i, flavour, mask_type, size, overlap, extend, centre, dict(zip(vals.tolist(), cnt.tolist())))
                    i += 1
    out["cases"] = np.array(meta, dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "labels_extend.npz")
    np.savez_compressed(path, **out)
    print("oracle == reference on", i, "cases; wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
