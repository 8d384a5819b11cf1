#!/usr/bin/env python3
"""Golden vectors for the metadata planes (batch/dataset.py:288-351): the REFERENCE's get_crop_memmap run on a fake
Echogram carrying the three per-ping vectors, for crops in the interior and over every border of the echogram, with all
seven planes and with subsets.  Build container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.make_golden_tiling as g  # noqa: E402,F401  (stubs the optional imports, puts the reference on sys.path)
from tools.fake_reader import FakeEchogram, synth_survey  # noqa: E402
from oracle import tiling_oracle as orc  # noqa: E402
from batch.dataset import get_crop_memmap  # noqa: E402  (reference)


def main():
    sv, labels, seabed = synth_survey(n_pings=700, n_range=420, seed=11)
    eg = FakeEchogram(np.ascontiguousarray(sv.swapaxes(1, 2)), np.ascontiguousarray(labels.T), seabed,
                      frequencies=(18, 38, 120, 200))
    rng = np.random.Generator(np.random.PCG64(5))
    n = eg.shape[1]
    t0 = 737000.25
    eg.time_vector = t0 + np.cumsum(rng.uniform(5e-6, 9e-6, size=n))
    eg.portion_of_day_vector = eg.time_vector % 1
    eg.portion_of_year_scalar = 7 / 12 + 19 / 366 + 6 / 366 / 24
    eg.time_vector_diff = np.concatenate((np.diff(eg.time_vector), [eg.time_vector[-1] - eg.time_vector[-2]])) / 6e-6 - 1
    all_on = {k: True for k in orc.META_KEYS}
    subset = dict(all_on, portion_day=False, depth_rel=False)
    centres = np.array([[200, 350], [128, 128], [40, 10], [400, 690], [210, -60], [300, 760], [127, 699], [5, 0]])
    out = {}
    for name, mc in (("all", all_on), ("subset", subset)):
        planes = []
        for c in centres:
            _, meta, _ = get_crop_memmap(eg, np.array(c), [256, 256], [18, 38, 120, 200], mc)
            ref = np.asarray(meta, dtype=np.float64)
            mine = orc.meta_planes(c, (256, 256), mc, eg.portion_of_year_scalar, eg.portion_of_day_vector,
                                   eg.time_vector_diff, eg._seabed)
            assert ref.shape == mine.shape and np.array_equal(ref, mine, equal_nan=True), (name, c)
            planes.append(ref.astype(np.float32))
        out["planes_" + name] = np.stack(planes)
        print(name, out["planes_" + name].shape, "oracle == reference (bit for bit, float64)")
    # a water column not deeper than the window: the centre row is moved to the middle (dataset.py:256-258)
    eg2 = FakeEchogram(eg.sv[:, :200], eg.labels[:200], np.minimum(seabed, 190), frequencies=(18, 38, 120, 200))
    for k in ("portion_of_day_vector", "portion_of_year_scalar", "time_vector_diff"):
        setattr(eg2, k, getattr(eg, k))
    c = np.array([30, 300])
    _, meta, _ = get_crop_memmap(eg2, c, [256, 256], [18, 38, 120, 200], all_on)
    assert c[0] == 100                                   # (adjusted in place by the reference)
    mine = orc.meta_planes(c, (256, 256), all_on, eg2.portion_of_year_scalar, eg2.portion_of_day_vector,
                           eg2.time_vector_diff, eg2._seabed)
    assert np.array_equal(np.asarray(meta, dtype=np.float64), mine, equal_nan=True)
    path = os.path.join(ROOT, "tests", "golden", "meta_planes.npz")
    np.savez_compressed(path, centres=centres, seabed=np.asarray(eg._seabed), portion_day=eg.portion_of_day_vector,
                        time_diff=eg.time_vector_diff, portion_year=eg.portion_of_year_scalar,
                        shallow_centre=c, shallow_planes=np.asarray(meta, dtype=np.float32),
                        shallow_seabed=np.asarray(eg2._seabed), **out)
    print("saved", os.path.getsize(path))


if __name__ == "__main__":
    main()
