#!/usr/bin/env python3
"""Golden vectors for the seabed-MASK rule of tiled inference (mask_label_seabed.py:47-49 reads the reader's 2-D mask,
not the seabed vector): the REFERENCE's DatasetGriddedReader preload path + label / data transforms + fill_out_array
on a fake reader whose stored mask has pings without a detected bottom (all-zero columns) and pings with holes --
cases where ``range >= seabed[ping]`` (seabed = argmax of the mask, data_reader.py:864-865) is NOT the mask.
Build container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.make_golden_tiling as g  # noqa: E402  (stubs the optional imports, puts the reference on sys.path)
from tools.fake_reader import FakeZarrReader, holey_seabed_mask, linear_predictor, synth_survey  # noqa: E402
from oracle import tiling_oracle as orc  # noqa: E402
from batch.dataset import DatasetGriddedReader  # noqa: E402  (reference)
from batch.transforms import define_data_transform, define_label_transform_test  # noqa: E402


def main():
    try:
        from pipeline_train_predict.save_predict import fill_out_array
    except Exception:  # noqa: BLE001
        fill_out_array = g.ref_fill_out_array
    sv, labels, seabed = synth_survey()
    sv, labels, seabed = sv[:, :500], labels[:500], seabed[:500]
    mask = holey_seabed_mask(seabed, sv.shape[2])
    reader = FakeZarrReader(sv, labels, seabed, mask=mask)
    n_pings, n_range = reader.shape
    freqs = [18000, 38000, 120000, 200000]
    patch, overlap = [256, 256], 20
    ds = DatasetGriddedReader(reader, patch, freqs, meta_channels=[], grid_start=0, grid_end=n_pings,
                              patch_overlap=overlap, data_preload=True, augmentation_function=None,
                              label_transform_function=define_label_transform_test(freqs, label_masks="all",
                                                                                   patch_overlap=overlap),
                              data_transform_function=define_data_transform(False), grid_mode="all")
    out = np.zeros([2, n_range, n_pings])
    for i in range(len(ds)):
        item = ds[i]
        fill_out_array(out, linear_predictor(item["data"]), item["labels"], item["center_coordinates"], 0)
    o_out, _ = orc.predict_chunk(sv, labels, reader.seabed, 0, n_pings, linear_predictor, patch, overlap, seabed_mask=mask)
    v_out, _ = orc.predict_chunk(sv, labels, reader.seabed, 0, n_pings, linear_predictor, patch, overlap)
    print(f"{len(ds)} patches; oracle(mask) vs reference max abs diff {np.abs(o_out - out).max():.2e}; pixels where the "
          f"vector rule differs from the reference: {int(((v_out[0] != 0) != (out[0] != 0)).sum())}")
    assert np.abs(o_out - out).max() < 1e-6
    assert ((v_out[0] != 0) != (out[0] != 0)).sum() > 0
    path = os.path.join(ROOT, "tests", "golden", "tiling_mask.npz")
    np.savez_compressed(path, out_f16=out.astype(np.float16), n_pings=n_pings, n_range=n_range, overlap=overlap)
    print("saved", os.path.getsize(path))


if __name__ == "__main__":
    main()
