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


# ---- test-time chain (validation / evaluate flows): define_label_transform_test --------------------------------------
def _test_cases(golden_dir):
    fix = np.load(os.path.join(golden_dir, "labels_test.npz"))
    sv03, labels, seabed, mask = fix["sv03"], fix["labels"], fix["seabed"], fix["holey_mask"]
    sv = np.zeros((4,) + sv03.shape[1:], dtype=np.float32)
    sv[0], sv[3] = sv03[0], sv03[1]
    from oracle import tiling_oracle as torc
    for i, (flav, size, overlap, cy, cx) in enumerate(fix["cases"].tolist()):
        lab = torc.crop(np.ascontiguousarray(labels.T), (cy, cx), (size, size), -100).astype(np.int64)
        data = torc.crop(np.ascontiguousarray(sv.transpose(0, 2, 1)), (cy, cx), (size, size), 0).astype(np.float32)
        yield i, ("zarr", "zarrmask", "memm")[flav], size, overlap, (cy, cx), data, lab, seabed, mask, fix[f"c{i}/final"]


def test_oracle_matches_reference_test_label_transform(golden_dir):
    """oracle/labels_oracle.test_label_transform == the reference's define_label_transform_test + remove_nan_inf on 63
    crops: zarr reader (vector-expressible mask), zarr reader with a holey stored mask, memmap Echogram; three patch
    sizes / overlaps; centres at the survey corners, across the seabed, below the water column."""
    seen = set()
    n_range = None
    for i, flav, size, overlap, centre, data, lab, seabed, mask, final in _test_cases(golden_dir):
        n_range = mask.shape[1]
        got = orc.test_label_transform(data, lab, centre, 3, seabed, n_range, overlap,
                                       "memm" if flav == "memm" else "zarr", mask if flav == "zarrmask" else None)
        assert np.array_equal(got, final), f"case {i} ({flav}, {size}, {overlap}, {centre}): {np.argwhere(got != final)[:5]}"
        seen |= set(np.unique(final).tolist())
    assert seen == {-100, -70, -50, -30, -10, 0, 1, 2}


@pytest.mark.gpu
def test_labels_test_transform_kernel_matches_reference_golden(golden_dir):
    """crimac_labels_test_transform (batched: every case of one size in ONE launch) is bit-exact against the reference."""
    import torch
    from crimac_classifiers_unet_amd.hip import call, ptr
    cases = list(_test_cases(golden_dir))
    mask = cases[0][8]
    n_pings, n_range = mask.shape
    for flav in ("zarr", "zarrmask", "memm"):
        for size in (64, 96, 128):
            grp = [c for c in cases if c[1] == flav and c[2] == size]
            overlap = grp[0][3]
            data = torch.from_numpy(np.stack([c[5] for c in grp])).cuda()
            cen = torch.tensor([c[4] for c in grp], dtype=torch.int64).cuda()
            sb = torch.from_numpy(grp[0][7].astype(np.int32)).cuda()
            mk = torch.from_numpy(np.ascontiguousarray(mask)).cuda() if flav == "zarrmask" else None
            for dt in (torch.int16, torch.int64):
                lab = torch.from_numpy(np.stack([c[6] for c in grp])).to(dt).cuda()
                out = torch.full((len(grp), size, size), 7, dtype=torch.int16, device="cuda")
                call("crimac_labels_test_transform", ptr(lab), lab.element_size(), ptr(data), 3, 1e-7, 1e-4, ptr(cen),
                     None if mk is not None else ptr(sb), 0, n_pings, ptr(mk), 0, n_pings if mk is not None else 0, n_range, 10,
                     1 if flav == "memm" else 0, overlap, ptr(out), len(grp), 4, size, size)
                torch.cuda.synchronize()
                for k, c in enumerate(grp):
                    assert np.array_equal(out[k].cpu().numpy(), c[9]), (flav, size, c[4], dt)


@pytest.mark.gpu
def test_validation_histograms_from_raw_crops_equal_the_host_transformed_batches(golden_dir):
    """SegPipe.use_gpu_test_transform: a validation DataLoader that hands RAW crops (linear sv, raw annotation ids, centre
    coordinates) gives the same PR histograms and loss as one that hands the REFERENCE-transformed labels (the golden
    `final` arrays, i.e. define_label_transform_test + remove_nan_inf as the reference ran them) and dB data."""
    import torch
    import yaml
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import synth
    from oracle import tiling_oracle as torc
    from tools.fake_reader import FakeZarrReader
    fix = np.load(os.path.join(golden_dir, "labels_test.npz"))
    sv = np.zeros((4,) + fix["sv03"].shape[1:], dtype=np.float32)
    sv[0], sv[3] = fix["sv03"][0], fix["sv03"][1]
    sv[1], sv[2] = sv[3] * 0.5, sv[0] * 2.0                     # (channels 1 / 2 are not in the fixture: any finite data)
    sv[1][~np.isfinite(sv[1])] = 1e-5
    sv[2][~np.isfinite(sv[2])] = 1e-5
    reader = FakeZarrReader(sv, fix["labels"].astype(np.int64), fix["seabed"].astype(np.int64))
    cases = [c for c in _test_cases(golden_dir) if c[1] == "zarr" and c[2] == 96]
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(pkg.__file__), "configs", "pipeline_config.yaml")))
    cfg.update(save_model_params=False, data_mode="zarr")
    pipe = pkg.SegPipeUNet(experiment_name="t", **cfg)
    pipe.model.load_state_dict(synth.synth_state_dict(seed=0))
    pipe.model.to(pipe.device)
    crit = pipe.get_criterion()
    raw, host = [], []
    for k in range(0, len(cases) - 1, 2):
        grp = cases[k:k + 2]
        lin = np.stack([torc.crop(np.ascontiguousarray(sv.transpose(0, 2, 1)), c[4], (96, 96), 0).astype(np.float32) for c in grp])
        cen = torch.tensor([c[4] for c in grp], dtype=torch.int64)
        raw.append({"data": torch.from_numpy(lin), "labels": torch.from_numpy(np.stack([c[6] for c in grp]).astype(np.int16)),
                    "center_coordinates": cen})
        db = np.stack([torc.data_transform(x)[0] for x in lin])
        host.append({"data": torch.from_numpy(db), "labels": torch.from_numpy(np.stack([c[9] for c in grp])),
                     "center_coordinates": cen})
    hp0, hn0, loss0 = pipe.get_pr_histograms_dataloader(host, criterion=crit)
    pipe.use_gpu_test_transform(reader, patch_overlap=20)
    hp1, hn1, loss1 = pipe.get_pr_histograms_dataloader(raw, criterion=crit)
    # labels decide which pixels are counted and on which side: those totals are exact.  The probabilities behind the
    # bins come from two dB transforms (numpy log10 on the host, the GPU's in the raw flow) that may differ in the last
    # bit, so a few pixels may land in a neighbouring float16 bin
    assert hp0.sum() + hn0.sum() > 1000
    assert hp0.sum() == hp1.sum() and hn0.sum() == hn1.sum()
    moved = np.abs(np.cumsum(hp0) - np.cumsum(hp1)).sum() + np.abs(np.cumsum(hn0) - np.cumsum(hn1)).sum()
    assert moved <= 0.01 * (hp0.sum() + hn0.sum()), moved
    f0 = pipe.compute_evaluation_metrics_from_histograms(hp0, hn0)["F1"].max()
    f1 = pipe.compute_evaluation_metrics_from_histograms(hp1, hn1)["F1"].max()
    assert abs(f0 - f1) <= 1e-3 * max(f0, 1e-6)
    assert abs(loss0 - loss1) <= 1e-5 * abs(loss0)
    pipe.use_gpu_test_transform(None)
    assert pipe._test_source is None


# ---- eval_mode 'region' / 'trace': get_extended_label_mask_for_crop ------------------------------------------------------
def _extend_cases(golden_dir):
    base = np.load(os.path.join(golden_dir, "labels_test.npz"))
    fix = np.load(os.path.join(golden_dir, "labels_extend.npz"))
    sv = np.zeros((4,) + base["sv03"].shape[1:], dtype=np.float32)
    sv[0], sv[3] = base["sv03"][0], base["sv03"][1]
    labels, seabed = base["labels"], base["seabed"]
    n_pings, n_range = labels.shape
    from oracle import tiling_oracle as torc
    for i, (flav, mt, size, overlap, extend, cy, cx) in enumerate(fix["cases"].tolist()):
        lab = torc.crop(np.ascontiguousarray(labels.T), (cy, cx), (size, size), -100).astype(np.int64)
        data = torc.crop(np.ascontiguousarray(sv.transpose(0, 2, 1)), (cy, cx), (size, size), 0).astype(np.float32)
        flavour, mask_type = ("zarr", "memm")[flav], ("region", "trace")[mt]
        shape0 = n_pings if flavour == "zarr" else n_range          # `echogram.shape[0]` of the two reader kinds
        yield (i, flavour, mask_type, size, overlap, extend, (cy, cx), data, lab, seabed, n_range,
               orc.extend_boxes(fix["boxes"], mask_type, extend, shape0), fix[f"c{i}/final"])


def test_oracle_matches_reference_extended_label_mask(golden_dir):
    """define_label_transform_test(label_masks='region' | 'trace') + remove_nan_inf as the imported reference ran them
    (tools/make_golden_labels_extend.py): 48 crops, two reader kinds, two extend sizes."""
    n, seen = 0, set()
    for i, flav, mt, size, overlap, extend, centre, data, lab, seabed, n_range, boxes, final in _extend_cases(golden_dir):
        got = orc.test_label_transform(data, lab, centre, 3, seabed, n_range, overlap,
                                       "memm" if flav == "memm" else "zarr", boxes_extended=boxes)
        assert np.array_equal(got, final), (i, flav, mt, size, centre)
        seen |= set(np.unique(final).tolist())
        n += 1
    assert n == 48 and -1 in seen and 1 in seen and -100 in seen


@pytest.mark.gpu
def test_labels_extend_mask_kernel_matches_reference_golden(golden_dir):
    """crimac_labels_test_transform + crimac_labels_extend_mask, batched per (reader kind, mask type, size), bit-exact."""
    import torch
    from crimac_classifiers_unet_amd.hip import call, ptr
    cases = list(_extend_cases(golden_dir))
    n_pings = np.load(os.path.join(golden_dir, "labels_test.npz"))["labels"].shape[0]
    groups = {}
    for c in cases:
        groups.setdefault((c[1], c[2], c[3]), []).append(c)
    assert len(groups) == 8
    for (flav, mt, size), grp in groups.items():
        overlap, n_range = grp[0][4], grp[0][10]
        data = torch.from_numpy(np.stack([c[7] for c in grp])).cuda()
        lab = torch.from_numpy(np.stack([c[8] for c in grp])).cuda()
        cen = torch.tensor([c[6] for c in grp], dtype=torch.int64).cuda()
        sb = torch.from_numpy(grp[0][9].astype(np.int32)).cuda()
        boxes = torch.from_numpy(np.ascontiguousarray(grp[0][11].astype(np.int32))).cuda()
        out = torch.full((len(grp), size, size), 7, dtype=torch.int16, device="cuda")
        call("crimac_labels_test_transform", ptr(lab), 8, ptr(data), 3, 1e-7, 1e-4, ptr(cen), ptr(sb), 0, n_pings, None, 0, 0,
             n_range, 10, 1 if flav == "memm" else 0, overlap, ptr(out), len(grp), 4, size, size)
        call("crimac_labels_extend_mask", ptr(out), ptr(data), 4, ptr(cen), ptr(boxes), int(boxes.shape[0]), -1, len(grp),
             size, size)
        torch.cuda.synchronize()
        for k, c in enumerate(grp):
            assert np.array_equal(out[k].cpu().numpy(), c[12]), (flav, mt, size, c[6])
    # more boxes than one filtering round holds (1024), most of them far away; and no boxes at all -> everything ignored
    c = groups[("zarr", "region", 128)][4]
    far = np.tile(np.array([[5000, 5010, 7000, 7040]], dtype=np.int32), (3000, 1))
    many = torch.from_numpy(np.ascontiguousarray(np.concatenate([far[:1500], c[11].astype(np.int32), far[1500:]]))).cuda()
    data, lab = torch.from_numpy(c[7][None]).cuda(), torch.from_numpy(c[8][None]).cuda()
    cen, sb = torch.tensor([c[6]], dtype=torch.int64).cuda(), torch.from_numpy(c[9].astype(np.int32)).cuda()
    for bx, want in ((many, c[12]), (None, None)):
        out = torch.empty((1, 128, 128), dtype=torch.int16, device="cuda")
        call("crimac_labels_test_transform", ptr(lab), 8, ptr(data), 3, 1e-7, 1e-4, ptr(cen), ptr(sb), 0, n_pings, None, 0, 0,
             c[10], 10, 0, c[4], ptr(out), 1, 4, 128, 128)
        call("crimac_labels_extend_mask", ptr(out), ptr(data), 4, ptr(cen), ptr(bx), 0 if bx is None else int(bx.shape[0]), -1,
             1, 128, 128)
        torch.cuda.synchronize()
        got = out[0].cpu().numpy()
        if want is None:
            assert set(np.unique(got).tolist()) <= {-1, -100} and (got == -100).sum() == (~np.isfinite(c[7][0])).sum()
        else:
            assert np.array_equal(got, want)


def _raw_pipe(golden_dir, reader, **over):
    import yaml
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import synth
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(pkg.__file__), "configs", "pipeline_config.yaml")))
    cfg.update(save_model_params=False, data_mode="zarr")
    cfg.update(over)
    pipe = pkg.SegPipeUNet(experiment_name="t", **cfg)
    pipe.model.load_state_dict(synth.synth_state_dict(seed=0))
    pipe.model.to(pipe.device)
    return pipe


def _full_sv(base):
    sv = np.zeros((4,) + base["sv03"].shape[1:], dtype=np.float32)
    sv[0], sv[3] = base["sv03"][0], base["sv03"][1]
    sv[1], sv[2] = sv[3] * 0.5, sv[0] * 2.0
    sv[1][~np.isfinite(sv[1])] = 1e-5
    sv[2][~np.isfinite(sv[2])] = 1e-5
    return sv


@pytest.mark.gpu
def test_raw_validation_batches_take_the_reader_mask_per_patch_where_the_vector_fails(golden_dir):
    """use_gpu_test_transform reads and checks the reader's seabed mask lazily, block by block: batches over
    blocks the seabed vector reproduces use the vector; a batch touching the holey part of a stored mask gets the
    reader's mask per patch -- labels bit-exact against the reference's in both cases."""
    import torch
    from oracle import tiling_oracle as torc
    from tools.fake_reader import FakeZarrReader
    base = np.load(os.path.join(golden_dir, "labels_test.npz"))
    sv = _full_sv(base)
    reader = FakeZarrReader(sv, base["labels"].astype(np.int64), base["seabed"].astype(np.int64), mask=base["holey_mask"])
    asked = []
    inner = reader.get_seabed_mask
    reader.get_seabed_mask = lambda s, n, *a, **k: (asked.append((int(s), int(n))), inner(s, n, *a, **k))[1]
    pipe = _raw_pipe(golden_dir, reader)
    pipe.TEST_SEABED_BLOCK = 16
    pipe.use_gpu_test_transform(reader, patch_overlap=20)
    assert asked == []                                          # nothing of the 2-D mask is read at set-up
    paths = []
    inner_sb = pipe._test_seabed
    pipe._test_seabed = lambda *a: (lambda r: (paths.append("vector" if r[0] is not None else "mask"), r)[1])(inner_sb(*a))
    cases = [c for c in _test_cases(golden_dir) if c[1] == "zarrmask" and c[2] == 96]
    for c in cases:                                             # one patch per batch: both paths get their batches
        lin = torc.crop(np.ascontiguousarray(sv.transpose(0, 2, 1)), c[4], (96, 96), 0).astype(np.float32)[None]
        batch = {"data": torch.from_numpy(lin), "labels": torch.from_numpy(c[6][None].astype(np.int16)),
                 "center_coordinates": torch.tensor([c[4]], dtype=torch.int64)}
        _, lab = pipe._predict_raw_batch(batch)
        s_, e_ = max(c[4][1] - 47, 0), min(c[4][1] + 49, 520)
        # fake_reader.holey_seabed_mask: pings 100-130 have NO bottom (the vector expresses that as seabed = n_range),
        # pings 300-340 a hole below the first seabed rows (no vector can)
        if s_ < 341 and e_ > 300:
            assert paths[-1] == "mask", (c[4], paths[-1])
        assert np.array_equal(lab[0].cpu().numpy(), c[9]), (c[4], paths[-1])
    assert set(paths) == {"vector", "mask"}, paths
    blocks = pipe._test_source["blocks"]
    assert not all(blocks.values()) and any(blocks.values())
    assert max(n for _, n in asked) <= 96                       # never more than a block / a patch span at a time


@pytest.mark.gpu
def test_raw_validation_batches_with_eval_mode_region_and_trace(golden_dir):
    """SegPipe(eval_mode='region' | 'trace').use_gpu_test_transform: the extended label mask on the GPU, labels bit-exact
    against the reference's define_label_transform_test(label_masks=eval_mode); a reader without bounding boxes raises
    (the reference fails there with an AttributeError), an unknown eval_mode raises."""
    import torch
    from tools.fake_reader import FakeZarrReader
    base = np.load(os.path.join(golden_dir, "labels_test.npz"))
    fix = np.load(os.path.join(golden_dir, "labels_extend.npz"))
    sv = _full_sv(base)
    labels, seabed = base["labels"].astype(np.int64), base["seabed"].astype(np.int64)
    plain = FakeZarrReader(sv, labels, seabed)
    boxed = FakeZarrReader(sv, labels, seabed, boxes=fix["boxes"])
    for mode in ("region", "trace"):
        pipe = _raw_pipe(golden_dir, boxed, eval_mode=mode)
        with pytest.raises(NotImplementedError, match="get_object_bounding_boxes"):
            pipe.use_gpu_test_transform(plain, patch_overlap=12, extend_size=7)
        assert pipe._test_source is None
        cases = [c for c in _extend_cases(golden_dir) if c[1] == "zarr" and c[2] == mode and c[3] == 128]
        pipe.use_gpu_test_transform(boxed, patch_overlap=12, extend_size=7)
        batch = {"data": torch.from_numpy(np.stack([c[7] for c in cases])),
                 "labels": torch.from_numpy(np.stack([c[8] for c in cases]).astype(np.int16)),
                 "center_coordinates": torch.tensor([c[6] for c in cases], dtype=torch.int64)}
        logits, lab = pipe._predict_raw_batch(batch)
        assert tuple(logits.shape) == (len(cases), 3, 128, 128) and bool(torch.isfinite(logits).all())
        for j, c in enumerate(cases):
            assert np.array_equal(lab[j].cpu().numpy(), c[12]), (mode, c[6])
        # the -1 pixels are counted as negatives by the metric, like the reference's select_valid_predictions keeps them
        hp, hn, _ = pipe.get_pr_histograms_dataloader([batch])
        valid = sum(int(np.isin(c[12], (-1, 0, 1, 2, -50)).sum()) for c in cases)
        assert hp.sum() + hn.sum() == valid and hp.sum() == sum(int((c[12] == 1).sum()) for c in cases)
    pipe = _raw_pipe(golden_dir, boxed, eval_mode="fish")
    with pytest.raises(ValueError, match="eval_mode"):
        pipe.use_gpu_test_transform(boxed)
