"""CPU: the C-ABI library loads and exports every symbol include/crimac_unet_hip.h declares (no
compute calls without a GPU), and host-side logic of the drop-in surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import yaml

import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import build, hip, parallel, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "crimac_unet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crimac_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    build.build()
    lib = ctypes.CDLL(build.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    # and the binding table covers the header exactly (plus version / last_error)
    assert set(hip.SIGNATURES) | {"crimac_version", "crimac_last_error", "crimac_wgrad_splits", "crimac_layer_desc_size",
                                  "crimac_wgrad_group_plan", "crimac_wgrad_group_layer_size"} == set(syms)
    # the split planner is a host-only query (no GPU): the level-0 shape fills one resident round
    lib2 = hip.load_library()
    assert lib2.crimac_wgrad_splits(0, 0, 64, 64, 32, 256, 256, 0) == 256        # bf16: one 8-wave workgroup per CU
    assert lib2.crimac_wgrad_splits(1, 0, 64, 64, 32, 256, 256, 0) == 512        # fp32 planes: two 4-wave workgroups
    assert lib2.crimac_wgrad_splits(0, 0, 1024, 1024, 32, 16, 16, 0) >= 1
    assert lib2.crimac_wgrad_splits(0, 2, 64, 64, 32, 256, 256, 0) < 0          # bad mode
    assert lib2.crimac_wgrad_splits(9, 0, 64, 64, 32, 256, 256, 0) < 0          # bad precision
    # ABI identity: the binding refuses a library of another version or with another descriptor layout
    header = open(os.path.join(ROOT, "include", "crimac_unet_hip.h")).read()
    assert int(re.search(r"#define CRIMAC_ABI_VERSION (\d+)", header).group(1)) == hip.ABI_VERSION
    assert lib2.crimac_version() == hip.ABI_VERSION
    assert lib2.crimac_layer_desc_size() == ctypes.sizeof(hip.LayerDesc)
    assert lib2.crimac_wgrad_group_layer_size() == ctypes.sizeof(hip.WgradGroupLayer)


def test_grouped_weight_gradient_plan_covers_every_item_exactly_once():
    """crimac_wgrad_group_plan (host only): for the encoder shapes of the benchmark and for ragged tiny shapes, every
    (layer, channel-tile pair, pixel split) appears in exactly one of the 8 queues; whole rounds of 8 splits put split s
    in queue s % 8 (the XCD that keeps its pixel range in L2); longest items come first."""
    lib = hip.load_library()
    cases = [(32, [(64, 64, 256, 256), (128, 64, 128, 128), (128, 128, 128, 128), (256, 128, 64, 64), (1024, 512, 16, 16)]),
             (2, [(64, 64, 32, 48), (128, 128, 16, 24), (512, 256, 4, 6)]), (1, [(64, 128, 8, 16)])]
    for B, shapes in cases:
        arr = (hip.WgradGroupLayer * len(shapes))()
        for d, (cf, cs, h, w) in zip(arr, shapes):
            d.CF, d.CS, d.f_ld, d.s_ld, d.Hf, d.Wf = cf, cs, cf, cs, h, w
        counts = (ctypes.c_int * 8)()
        cap = lib.crimac_wgrad_group_plan(0, arr, len(shapes), B, 0, None, 0, counts)
        assert cap > 0 and max(counts) == cap
        items = (ctypes.c_int * (8 * cap * 2))()
        assert lib.crimac_wgrad_group_plan(0, arr, len(shapes), B, 0, items, cap, counts) == cap
        seen = set()
        for x in range(8):
            last_cost = None
            for k in range(counts[x]):
                li, qs = items[2 * (x * cap + k)], items[2 * (x * cap + k) + 1]
                qt, sp = qs >> 16, qs & 0xFFFF
                d = arr[li]
                assert 0 <= sp < d.nsplits and 0 <= qt < ((d.CF + 63) // 64) * ((d.CS + 63) // 64)
                if d.nsplits % 8 == 0:
                    assert sp % 8 == x
                assert (li, qt, sp) not in seen
                seen.add((li, qt, sp))
                assert last_cost is None or d.tiles_per_block <= last_cost
                last_cost = d.tiles_per_block
        total = sum(d.nsplits * ((d.CF + 63) // 64) * ((d.CS + 63) // 64) for d in arr)
        assert len(seen) == total == sum(counts)
        for d in arr:
            assert d.ntiles == B * d.tiles_y * d.tiles_x and d.tiles_per_block * d.nsplits >= d.ntiles
            assert d.tiles_per_block * (d.nsplits - 1) < d.ntiles          # no empty split
    # the first layer (4 channels padded to 16) is refused: it keeps its own launch
    arr = (hip.WgradGroupLayer * 1)()
    arr[0].CF, arr[0].CS, arr[0].f_ld, arr[0].s_ld, arr[0].Hf, arr[0].Wf = 64, 16, 64, 16, 64, 64
    assert lib.crimac_wgrad_group_plan(0, arr, 1, 2, 0, None, 0, (ctypes.c_int * 8)()) < 0


def test_binding_refuses_a_library_with_another_abi_version(monkeypatch):
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "ABI_VERSION", hip.ABI_VERSION + 1)
    with pytest.raises(hip.HipLibraryError, match="ABI version"):
        hip.load_library()


def test_argument_validation_runs_without_gpu():
    """Argument checks precede any HIP call, so they are testable on the CPU box."""
    lib = hip.load_library()
    rc = lib.crimac_conv3x3(0, None, 64, 1, 8, 8, 24, 64, None, None, None, None, 64, 0, 0, None, None, 1,
                            None, 0, None, 0, None)
    assert rc < 0 and b"Cin" in lib.crimac_last_error()
    rc = lib.crimac_sgd_momentum(None, None, None, 0, 0.1, 0.9, 1.0, 0, None)
    assert rc < 0


def test_module_surface_matches_reference_layout():
    m = pkg.UNet_Baseline(n_classes=3, in_channels=4)
    sd = m.state_dict()
    shapes = synth.unet_state_shapes()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == shapes[k] for k in sd)
    assert sum(p.numel() for p in m.parameters()) == 31044227
    m.load_state_dict(synth.synth_state_dict())          # reference-style checkpoint loads
    with pytest.raises(ValueError):
        pkg.UNet_Baseline(3, 4, up_mode="bogus")
    with pytest.raises(ValueError):
        pkg.UNet_Baseline(3, 4, merge_mode="bogus")
    with pytest.raises(NotImplementedError):
        pkg.UNet_Baseline(3, 4, merge_mode="add")


def test_same_seed_gives_reference_style_init():
    torch.manual_seed(10)
    a = pkg.UNet_Baseline(3, 4).state_dict()
    torch.manual_seed(10)
    b = pkg.UNet_Baseline(3, 4).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    w = a["down_convs.1.main.0.weight"]
    assert abs(float(w.abs().max()) - 1 / (64 * 9) ** 0.5) < 1e-3


def test_segpipe_accepts_the_reference_yaml_keys():
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(pkg.__file__), "configs", "pipeline_config.yaml")))
    cfg["save_model_params"] = False
    pipe = pkg.SegPipeUNet(experiment_name="cpu", **cfg)
    assert pipe.frequencies == [18, 38, 120, 200] and pipe.window_size == [256, 256]
    assert pipe.model.in_channels == 4 and pipe.model_name == "cpu"
    crit = pipe.get_criterion()
    assert crit.weight.tolist() == [10.0, 300.0, 250.0]
    lab = torch.tensor([-100, -70, -50, -30, -10, 0, 1, 2])
    assert pipe.set_label_ignore_val(lab.clone()).tolist() == [-100, -100, 0, -100, -100, 0, 1, 2]
    with pytest.raises(AssertionError):
        pkg.SegPipeUNet(checkpoint_dir=None, experiment_name="x", **{**cfg, "save_model_params": True})
    with pytest.raises(ValueError):
        pkg.SegPipeUNet(experiment_name="x", **{**cfg, "loss_type": "Focal"}).get_criterion()
    assert pkg.get_in_channels([]) == 0
    assert pkg.get_in_channels({"portion_year": True, "portion_day": True, "depth_rel": False,
                                "depth_abs_surface": True, "depth_abs_seabed": False,
                                "time_diff": False}) == 4


def test_cpu_forward_fails_loudly_not_silently():
    m = pkg.UNet_Baseline(3, 4)
    with pytest.raises(Exception, match="no CPU fallback"):
        m(torch.zeros(1, 4, 32, 32))
    with pytest.raises(hip.HipLibraryError):
        hip.ptr(torch.zeros(3))


def test_metric_helpers_f1():
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(pkg.__file__), "configs", "pipeline_config.yaml")))
    cfg["save_model_params"] = False
    pipe = pkg.SegPipeUNet(experiment_name="cpu", **cfg)
    labels = np.array([1, 1, 0, 0, 2, -100], dtype=np.int8)
    preds = np.array([0.9, 0.8, 0.1, 0.7, 0.2, 0.99], dtype=np.float16)
    l, p = pipe.select_valid_predictions(labels.copy(), preds)
    assert len(l) == 5
    m = pipe.compute_evaluation_metrics(l, p)
    assert abs(m["F1"].max() - 1.0) < 1e-6      # threshold 0.8 separates both sandeel pixels


def test_bucket_and_shard_plans():
    assert parallel.bucket_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert parallel.shard_indices(10, 1, 4) == [1, 5, 9]
    allidx = sorted(i for r in range(8) for i in parallel.shard_indices(95, r, 8))
    assert allidx == list(range(95))             # 95 patches of one 4096-ping chunk


def test_pr_report_csv_has_the_reference_dataframe_layout(tmp_path):
    """validate_model_testing's csv (reference pipeline.py:358-361 writes DataFrame(metrics).to_csv): index column,
    precision / recall / thresholds / F1, NaN threshold in the last row."""
    import csv
    import numpy as np
    from crimac_classifiers_unet_amd.pipeline import SegPipe, write_pr_csv
    hp = np.zeros(16384, dtype=np.int64)
    hn = np.zeros(16384, dtype=np.int64)
    bits = lambda v: int(np.float16(v).view(np.uint16))
    hp[bits(0.9)], hp[bits(0.6)], hn[bits(0.7)], hn[bits(0.1)] = 3, 1, 2, 10
    m = SegPipe.compute_evaluation_metrics_from_histograms(hp, hn)
    # same numbers as sklearn on the expanded vectors
    from sklearn.metrics import precision_recall_curve
    y = np.r_[np.ones(4), np.zeros(12)]
    sc = np.r_[[0.9] * 3, [0.6], [0.7] * 2, [0.1] * 10].astype(np.float16)
    p, r, t = precision_recall_curve(y, sc, pos_label=1)
    assert np.allclose(m["precision"], p) and np.allclose(m["recall"], r) and np.allclose(m["thresholds"], t)
    m["thresholds"] = np.append(m["thresholds"], np.nan)
    write_pr_csv(m, tmp_path / "pr.csv")
    rows = list(csv.reader(open(tmp_path / "pr.csv")))
    assert rows[0] == ["", "precision", "recall", "thresholds", "F1"]
    assert len(rows) == len(p) + 1 and rows[-1][3] == "" and rows[1][0] == "0"
    assert abs(float(rows[1][1]) - p[0]) < 1e-15


def test_evaluate_flow_needs_the_reference_data_stack_or_injected_factories():
    import inspect
    from crimac_classifiers_unet_amd import evaluate
    with pytest.raises(ImportError, match="dataset_cls"):
        evaluate.validate_model_survey_memm([], None, [], [256, 256], 20, "all", 4, 0, "/tmp", "/tmp")
    with pytest.raises(ImportError, match="label_transform_factory"):
        evaluate.validate_model_survey_zarr([1], None, [], [256, 256], 20, "all", 4, 0, "/tmp", "/tmp",
                                            dataset_cls=object, data_transform_factory=lambda m: None)
    # the product module never imports the reference package by itself: only from a path the caller names
    src = inspect.getsource(evaluate)
    assert "from batch" not in src and "import batch" not in src
    with pytest.raises(ImportError, match="host data stack"):
        evaluate.data_stack_factories("/nonexistent/path", memm=True)


def test_spawned_ranks_time_out_and_are_reaped():
    import sys
    from crimac_classifiers_unet_amd import launch
    rc, _ = launch.spawn_ranks([sys.executable, "-c", "import time; time.sleep(60)"], 2, timeout=2)
    assert rc != 0


def test_planes_argument_layout_matches_the_header():
    text = open(os.path.join(ROOT, "include", "crimac_unet_hip.h")).read()
    assert "#define CRIMAC_PLANES_FWD_FP16 16" in text and "#define CRIMAC_PLANES_DG_FP16 32" in text
    assert "#define CRIMAC_F32H3_WSHIFT 8" in text
    assert hip.PLANES_FP16 == 1 | 16 | 32 and hip.PLANES_F32H3 == 2 | 16 | (8 << 8)
    # ('h3f' is an engine mode: forward launches carry H3P, the backward pass's contractions CRIMAC_PREC_H3F_BWD)
    assert hip.PREC_NAMES == {"bf16": 0, "f32x3": 1, "f32x6": 2, "fp16": 3, "f32h3": 4, "h3p": 5, "h3f": 5}
    assert "#define CRIMAC_PREC_H3F_BWD 6" in text and hip.PREC_H3F_BWD == 6
    assert hip.PREC_BACKWARD[hip.PREC_F32H3] == hip.PREC_F32X3
    lib = hip.load_library()
    # argument checks of the packers run without a GPU: a plane count of 0 or stray bits are refused
    assert lib.crimac_pack_layers(None, 1, 1, None) < 0
    import ctypes
    d = (hip.LayerDesc * 1)()
    assert lib.crimac_pack_layers(ctypes.byref(d), 1, 0, None) < 0
    assert lib.crimac_pack_layers(ctypes.byref(d), 1, 2 | 64, None) < 0


class _F64Crops(torch.utils.data.Dataset):
    """What the reference's zarr Dataset hands over (batch/dataset.py:361): float64 data, int16 labels, int64 centres."""

    def __len__(self):
        return 12

    def __getitem__(self, i):
        rng = np.random.default_rng(i)
        return {"data": np.power(10.0, rng.uniform(-9, -1, (4, 16, 16))), "labels": rng.integers(-1, 3, (16, 16)).astype(np.int16),
                "center_coordinates": np.array([i, 2 * i], dtype=np.int64)}


@pytest.mark.parametrize("workers", [0, 2])
def test_collate_float32_gives_the_values_of_the_reference_cast(workers):
    """`collate_float32` (yaml key of the same name; staging.py): the training DataLoader's float64 crops are cast per sample in
    the collate -- the batch equals `default_collate(...)['data'].float()` (pipeline.py:163) bit for bit, the other keys are
    untouched, a DataLoader with a collate function of its own is left alone, and the early page release still recognises
    the batches as worker-collated."""
    from torch.utils.data import DataLoader, default_collate
    from crimac_classifiers_unet_amd import staging
    ds = _F64Crops()
    ref = list(DataLoader(ds, batch_size=4, shuffle=False, num_workers=0))
    dl = DataLoader(ds, batch_size=4, shuffle=False, num_workers=workers)
    assert staging.use_collate_float32(dl) and dl.collate_fn is staging.collate_float32
    assert staging.collated_in_worker(dl) == (workers > 0)
    got = list(dl)
    assert len(got) == len(ref) == 3
    for g, r in zip(got, ref):
        assert r["data"].dtype == torch.float64 and g["data"].dtype == torch.float32
        assert torch.equal(g["data"], r["data"].float())
        assert g["labels"].dtype == torch.int16 and torch.equal(g["labels"], r["labels"])
        assert torch.equal(g["center_coordinates"], r["center_coordinates"])
    own = DataLoader(ds, batch_size=4, collate_fn=lambda s: default_collate(s))
    assert not staging.use_collate_float32(own)
    assert not staging.use_collate_float32(DataLoader(ds, batch_size=None))           # no automatic batching: nothing to collate
    # samples that are not dicts, and float32 data, pass through default_collate unchanged
    t = staging.collate_float32([np.ones(3), np.zeros(3)])
    assert t.dtype == torch.float64 and t.shape == (2, 3)
    f = staging.collate_float32([{"data": np.ones((2, 2), np.float32)}, {"data": np.zeros((2, 2), np.float32)}])
    assert f["data"].dtype == torch.float32
