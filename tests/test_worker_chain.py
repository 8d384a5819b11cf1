"""The worker-side transform chain of the reference (batch/dataset.py:89-103) as restated in
oracle/augment_oracle.worker_train_chain -- the CPU baseline of bench.py's ``train_loop_raw`` leg -- against outputs of the
imported reference (tools/make_golden_worker_chain.py), and the raw-crop Dataset that feeds both legs."""
import os

import numpy as np
import pytest

from crimac_classifiers_unet_amd import synth
from oracle import augment_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "worker_chain.npz")


def test_worker_chain_oracle_reproduces_the_reference_bit_for_bit():
    fix = np.load(GOLD)
    cases = sorted({k.split("/")[0] for k in fix.files if k.endswith("/seed")})
    assert len(cases) >= 6
    n_f64 = 0
    for c in cases:
        seed = int(fix[f"{c}/seed"])
        for dtype in (np.float32, np.float64):
            tag = f"{c}_{np.dtype(dtype).name}"
            if f"{tag}/out_data" not in fix.files:
                continue
            n_f64 += dtype is np.float64
            d, lab = augment_oracle.worker_train_chain(fix[f"{c}/data"].astype(dtype), fix[f"{c}/labels"].copy(),
                                                       np.random.RandomState(seed))
            assert d.dtype == fix[f"{tag}/out_data"].dtype
            assert np.array_equal(d, fix[f"{tag}/out_data"]), tag
            assert np.array_equal(lab, fix[f"{tag}/out_labels"]), tag
            assert set(np.unique(lab)) <= {-100, 0, 1, 2}
    assert n_f64 >= 2


def test_raw_crop_geometry_and_boundary_values():
    """utils/np.py:378-380: patch pixel p <-> data coordinate centre - n // 2 + 1 + p; outside the survey data = 0 and
    labels = -100 (dataset.py:360-362); NaN -> 0 (dataset.py:402)."""
    r = synth.SyntheticSurveyReader(n_pings=300, n_range=200, block=300, schools=8, bad_frac=0.01, seed=3)
    for centre in ((100, 150), (3, 5), (199, 299), (0, 0), (120, 290)):
        data, labels = synth.raw_crop(r, centre, (64, 48), np.float64)
        assert data.shape == (4, 64, 48) and data.dtype == np.float64 and labels.dtype == np.int16
        for py in (0, 17, 63):
            for px in (0, 30, 47):
                y, x = centre[0] - 32 + 1 + py, centre[1] - 24 + 1 + px
                if 0 <= y < 200 and 0 <= x < 300:
                    want = r.sv[:, x, y].astype(np.float64)
                    want[np.isnan(want)] = 0
                    assert np.array_equal(data[:, py, px], want, equal_nan=True)
                    assert labels[py, px] == r.labels[x, y]
                else:
                    assert (data[:, py, px] == 0).all() and labels[py, px] == -100
        assert not np.isnan(data).any()


def test_raw_crop_dataset_is_a_function_of_seed_and_index():
    r = synth.SyntheticSurveyReader(n_pings=512, n_range=300, block=512, schools=10, seed=4)
    a, b = synth.RawCropDataset(r, (64, 64), 10, seed=1), synth.RawCropDataset(r, (64, 64), 10, seed=1)
    c = synth.RawCropDataset(r, (64, 64), 10, seed=2)
    assert len(a) == 10
    for i in (0, 3, 9):
        assert np.array_equal(a[i]["center_coordinates"], b[i]["center_coordinates"])
        assert np.array_equal(a[i]["data"], b[i]["data"])
        assert set(a[i]) == {"data", "labels", "center_coordinates"}
        assert a[i]["labels"].dtype == np.int16 and a[i]["center_coordinates"].dtype == np.int64
    assert any(not np.array_equal(a[i]["center_coordinates"], c[i]["center_coordinates"]) for i in range(10))
    seen = synth.RawCropDataset(r, (64, 64), 1, transform=lambda d, l, i: (d * 0 + 1, l * 0 + 2))[0]
    assert (seen["data"] == 1).all() and (seen["labels"] == 2).all()


def test_default_survey_is_unchanged_by_the_school_option():
    """The tiled leg's survey (schools = 0) must be the array it always was."""
    a = synth.SyntheticSurveyReader(n_pings=64, n_range=32, block=64, seed=1)
    rng = np.random.Generator(np.random.PCG64(1))
    assert np.array_equal(a.sv, np.power(10.0, rng.uniform(-7.5, 0.0, size=(4, 64, 32))).astype(np.float32))
    assert not a.labels.any()
