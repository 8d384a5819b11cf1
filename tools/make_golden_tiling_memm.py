#!/usr/bin/env python3
"""Golden vectors for the MEMM flavour of whole-echogram inference (save_reader_predictions_memm,
pipeline_train_predict/save_predict.py:222-265): the REFERENCE's own DatasetGriddedReader (non-preload path ->
get_crop_memmap), define_data_transform_test (remove_nan_inf, db_with_limits, set_data_border_value),
define_label_transform_test and fill_out_array, run on the fake in-memory Echogram.  Build container only."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/crimac_unet")
for name in ("dask", "xarray", "numcodecs", "tqdm"):
    if name in sys.modules:
        continue
    try:
        __import__(name)
    except Exception:
        m = types.ModuleType(name)
        if name == "dask":
            m.config = types.SimpleNamespace(set=lambda **kw: None)
        if name == "numcodecs":
            m.Blosc = object
        sys.modules[name] = m

from tools.fake_reader import FakeEchogram, synth_survey, linear_predictor  # noqa: E402
from tools.make_golden_tiling import ref_fill_out_array  # noqa: E402
from oracle import tiling_oracle as orc  # noqa: E402

from batch.dataset import DatasetGriddedReader  # noqa: E402  (reference)
from batch.transforms import define_data_transform_test, define_label_transform_test  # noqa: E402


def run(tag, n_pings, n_range, seed):
    sv, labels, seabed = synth_survey(n_pings=n_pings, n_range=n_range, seed=seed)
    sv_hw = np.ascontiguousarray(sv.swapaxes(1, 2))               # memm orientation [C, range, pings]
    labels_hw = np.ascontiguousarray(labels.T)
    eg = FakeEchogram(sv_hw, labels_hw, seabed)
    freqs, patch, overlap = [18, 38, 120, 200], [256, 256], 20
    ds = DatasetGriddedReader(eg, patch, freqs, meta_channels=[], grid_start=0, grid_end=n_pings,
                              patch_overlap=overlap, augmentation_function=None,
                              label_transform_function=define_label_transform_test(freqs, label_masks="all",
                                                                                   patch_overlap=overlap),
                              data_transform_function=define_data_transform_test(False), grid_mode="all")
    assert not ds.data_preload
    out = np.zeros([2, n_range, n_pings])
    keep = None
    centres = []
    for i in range(len(ds)):
        item = ds[i]
        preds = linear_predictor(item["data"]).astype(np.float16)          # save_predict.py:252
        ref_fill_out_array(out, preds, item["labels"], item["center_coordinates"], 0)
        centres.append(np.array(item["center_coordinates"]))
        if i == len(ds) // 2 or (keep is None and (item["labels"] == -100).any()):
            keep = (item["data"].astype(np.float32), item["labels"].astype(np.int16), np.array(item["center_coordinates"]))
    o_out = orc.predict_echogram_memm(sv_hw, labels_hw, seabed, linear_predictor, patch, overlap)
    err = np.abs(o_out - out).max()
    print(f"{tag}: {len(ds)} patches, oracle vs reference max abs diff {err:.2e}, written {np.mean(out[0] != 0):.3f}")
    assert err < 1e-6
    return {f"{tag}/out_f16": out.astype(np.float16), f"{tag}/centres": np.array(centres),
            f"{tag}/patch_data": keep[0], f"{tag}/patch_labels": keep[1], f"{tag}/patch_centre": keep[2],
            f"{tag}/shape": np.array([n_pings, n_range, seed])}


def main():
    fix = {}
    fix.update(run("deep", 900, 600, 11))
    fix.update(run("shallow", 700, 200, 12))        # water column not deeper than a patch: centre row = H // 2
    path = os.path.join(ROOT, "tests", "golden", "tiling_memm.npz")
    np.savez_compressed(path, **fix)
    print("saved", os.path.getsize(path))


if __name__ == "__main__":
    main()
