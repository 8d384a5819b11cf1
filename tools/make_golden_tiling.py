#!/usr/bin/env python3
"""Golden vectors for the tiled-inference plumbing, produced by the REFERENCE's own functions
(get_data_split, get_data_grid, DatasetGriddedReader preload path, define_label_transform_test,
define_data_transform, fill_out_array) run on the fake in-memory reader.  Build container only."""
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

from tools.fake_reader import FakeZarrReader, synth_survey, linear_predictor  # noqa: E402
from oracle import tiling_oracle as orc  # noqa: E402

from utils.preload_data_split import get_data_split  # noqa: E402  (reference)
from batch.samplers.gridded import get_data_grid  # noqa: E402
from batch.dataset import DatasetGriddedReader  # noqa: E402
from batch.transforms import define_data_transform, define_label_transform_test  # noqa: E402
import constants as C  # noqa: E402
from utils.np import patch_coord_to_data_coord  # noqa: E402


def ref_fill_out_array(out_array, preds, labels, center_coordinates, ping_start):
    """pipeline_train_predict/save_predict.py cannot be imported (torch DataLoader + xarray at module
    level are fine, but it pulls data.partition -> zarr); its fill_out_array body (save_predict.py:41-65)
    is exercised here through the same reference helpers it calls."""
    sel = np.argwhere((labels != C.LABEL_OVERLAP_VAL) & (labels != C.LABEL_SEABED_MASK_VAL)
                      & (labels != C.LABEL_BOUNDARY_VAL))
    if len(sel) == 0:
        return out_array
    y_label, x_label = np.transpose(sel)
    data_coords = patch_coord_to_data_coord(np.array(sel), np.array(center_coordinates), np.array(labels.shape))
    y_array, x_array = np.transpose(data_coords)
    x_array -= ping_start
    out_array[:, y_array, x_array] = preds[[C.SANDEEL, C.OTHER]][:, y_label, x_label]


def main():
    try:
        from pipeline_train_predict.save_predict import fill_out_array
        print("using reference fill_out_array")
    except Exception as e:  # noqa: BLE001
        print("save_predict not importable here (%s); using its helper-level restatement" % type(e).__name__)
        fill_out_array = ref_fill_out_array
    sv, labels, seabed = synth_survey()
    reader = FakeZarrReader(sv, labels, seabed)
    n_pings, n_range = reader.shape
    freqs = [18000, 38000, 120000, 200000]
    patch, overlap, preload = [256, 256], 20, 500
    splits = get_data_split([[0, n_pings]], preload)
    assert np.array_equal(splits, orc.get_data_split([[0, n_pings]], preload))
    data_transform = define_data_transform(False)
    label_transform = define_label_transform_test(freqs, label_masks="all", patch_overlap=overlap)
    outs, grids, first_patch = [], [], None
    for (s, e) in splits:
        ds = DatasetGriddedReader(reader, patch, freqs, meta_channels=[], grid_start=s, grid_end=e,
                                  patch_overlap=overlap, data_preload=True, augmentation_function=None,
                                  label_transform_function=label_transform,
                                  data_transform_function=data_transform, grid_mode="all")
        assert ds.data_preload
        grid = get_data_grid(reader, patch_size=patch, patch_overlap=overlap, start_ping=s, end_ping=e, mode="all")
        assert np.array_equal(grid, orc.get_data_grid(n_range, int(seabed[s:e].max()), s, e, patch, overlap))
        out = np.zeros([2, n_range, e - s])
        for i in range(len(ds)):
            item = ds[i]
            preds = linear_predictor(item["data"])
            fill_out_array(out, preds, item["labels"], item["center_coordinates"], s)
            if first_patch is None and i == len(ds) // 2:
                first_patch = (item["data"].astype(np.float32), item["labels"].astype(np.int16),
                               np.array(item["center_coordinates"]))
        o_out, o_grid = orc.predict_chunk(sv, labels, seabed, s, e, linear_predictor, patch, overlap)
        err = np.abs(o_out - out).max()
        print(f"chunk [{s},{e}): {len(grid)} patches, oracle vs reference max abs diff {err:.2e}, "
              f"written fraction {np.mean(out[0] != 0):.3f}")
        assert err < 1e-6
        outs.append(out)
        grids.append(grid)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tiling.npz"),
                        splits=splits, grid0=grids[0], grid_last=grids[-1],
                        out_f16=np.concatenate(outs, axis=2).astype(np.float16),
                        patch_data=first_patch[0], patch_labels=first_patch[1], patch_centre=first_patch[2],
                        n_pings=n_pings, n_range=n_range, preload=preload, overlap=overlap)
    print("saved", os.path.getsize(os.path.join(ROOT, "tests", "golden", "tiling.npz")))


if __name__ == "__main__":
    main()
