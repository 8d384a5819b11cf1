#!/usr/bin/env python3
"""Golden vectors + timing of the transform chain the REFERENCE runs per sample inside a DataLoader worker
(batch/dataset.py:89-103): define_data_augmentation -> define_label_transform_train -> define_data_transform
(batch/transforms.py:39-78), run by the imported reference on raw crops of the synthetic survey with numpy's global
generator seeded per case.  Build container only (imports /root/reference).

  * tests/golden/worker_chain.npz: small crops (inputs, seeds, the reference's outputs) -- pins
    oracle/augment_oracle.worker_train_chain bit for bit (tests/test_worker_chain.py);
  * with --time: per-stage milliseconds of the reference chain on 4 x 256 x 256 crops (float32 and float64), printed --
    the figure bench.py's host-chain baseline leg is compared with (profiles/r05_reference_host_chain.txt)."""
import os
import sys
import time
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

from batch.transforms import define_data_augmentation, define_label_transform_train, define_data_transform  # noqa: E402
from crimac_classifiers_unet_amd import synth  # noqa: E402
from oracle import augment_oracle  # noqa: E402

FREQS = [18, 38, 120, 200]


def reference_chain(data, labels, centre, echogram, stages=None):
    """Dataset.__getitem__ between get_crop and the return (dataset.py:89-105)."""
    aug, lt, dt = define_data_augmentation(), define_label_transform_train(FREQS), define_data_transform()
    t0 = time.perf_counter()
    data, labels, echogram = aug(data, labels, echogram)
    t1 = time.perf_counter()
    data, labels, _, echogram = lt(data, labels, centre, echogram)
    t2 = time.perf_counter()
    data, labels, echogram, _ = dt(data, labels, echogram, FREQS)
    t3 = time.perf_counter()
    if stages is not None:
        for k, v in (("augment", t1 - t0), ("label_transform", t2 - t1), ("data_transform", t3 - t2)):
            stages[k] = stages.get(k, 0.0) + v
    return data, labels.astype("int16")


def main():
    reader = synth.SyntheticSurveyReader(n_pings=4096, n_range=1024, block=4096, schools=60, bad_frac=1e-4, seed=5)
    echogram = types.SimpleNamespace(name="synthetic")
    out, worst = {}, 0.0
    ds = synth.RawCropDataset(reader, (64, 64), 64, seed=11)
    k = 0
    for i in range(64):
        item = ds[i]
        if k >= 6 or ((item["labels"] > 0).sum() == 0 and i % 8):       # mostly crops with schools in them
            continue
        for dtype in (np.float32, np.float64):
            data, labels = item["data"].astype(dtype), item["labels"].copy()
            np.random.seed(1000 + k)
            d_ref, l_ref = reference_chain(data.copy(), labels.copy(), item["center_coordinates"], echogram)
            d_orc, l_orc = augment_oracle.worker_train_chain(data.copy(), labels.copy(), np.random.RandomState(1000 + k))
            assert d_ref.dtype == d_orc.dtype, (d_ref.dtype, d_orc.dtype)
            worst = max(worst, float(np.abs(d_ref - d_orc).max()))
            assert np.array_equal(l_ref, l_orc), f"case {k}: labels differ"
            tag = f"c{k}_{np.dtype(dtype).name}"
            if dtype is np.float32 or k < 2:                     # (float64 outputs for two cases: fixture size)
                out[f"{tag}/out_data"], out[f"{tag}/out_labels"] = d_ref, l_ref
        out[f"c{k}/data"], out[f"c{k}/labels"], out[f"c{k}/seed"] = item["data"], item["labels"], np.int64(1000 + k)
        k += 1
    print(f"{k} cases; oracle vs reference: data max abs diff {worst}, labels identical")
    assert worst == 0.0
    path = os.path.join(ROOT, "tests", "golden", "worker_chain.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")

    if "--time" in sys.argv:
        ds = synth.RawCropDataset(reader, (256, 256), 400, seed=12)
        items = [ds[i] for i in range(60)]
        for dtype in (np.float32, np.float64):
            for name, fn in (("reference (imported)", None), ("oracle.worker_train_chain", augment_oracle.worker_train_chain)):
                stages, rs = {}, np.random.RandomState(7)
                np.random.seed(7)
                t0 = time.perf_counter()
                for it in items:
                    d, lab = it["data"].astype(dtype), it["labels"].copy()
                    if fn is None:
                        reference_chain(d, lab, it["center_coordinates"], echogram, stages)
                    else:
                        fn(d, lab, rs)
                dt_ms = 1e3 * (time.perf_counter() - t0) / len(items)
                st = ", ".join(f"{k_} {1e3 * v / len(items):.2f}" for k_, v in stages.items())
                print(f"{np.dtype(dtype).name:8s} {name:28s} {dt_ms:6.2f} ms per 4x256x256 crop ({1e3 / dt_ms:.0f} patches/s per core)"
                      + (f"  [{st}]" if st else ""))


if __name__ == "__main__":
    main()
