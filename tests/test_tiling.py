"""Tiled whole-survey inference (save_predict.py path).

CPU: the numpy oracle and the product's host planners against golden vectors produced by the
REFERENCE's own functions on the fake reader (tools/make_golden_tiling.py).
GPU: gather / scatter kernels and the whole chunk loop against the oracle and the same fixture.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from crimac_classifiers_unet_amd import tiled_inference as ti  # noqa: E402
from oracle import tiling_oracle as orc  # noqa: E402
from tools.fake_reader import FakeZarrReader, holey_seabed_mask, linear_predictor, synth_survey  # noqa: E402


@pytest.fixture(scope="module")
def fix(golden_dir):
    return np.load(os.path.join(golden_dir, "tiling.npz"))


@pytest.fixture(scope="module")
def survey():
    return synth_survey()


def test_oracle_matches_reference_golden(fix, survey):
    sv, labels, seabed = survey
    n_pings, n_range = int(fix["n_pings"]), int(fix["n_range"])
    splits = orc.get_data_split([[0, n_pings]], int(fix["preload"]))
    assert np.array_equal(splits, fix["splits"])
    outs = []
    for i, (s, e) in enumerate(splits):
        out, grid = orc.predict_chunk(sv, labels, seabed, s, e, linear_predictor, (256, 256), int(fix["overlap"]))
        if i == 0:
            assert np.array_equal(grid, fix["grid0"])
        if i == len(splits) - 1:
            assert np.array_equal(grid, fix["grid_last"])
        outs.append(out)
    full = np.concatenate(outs, axis=2).astype(np.float16)
    assert np.array_equal(full, fix["out_f16"])
    # one patch exactly as the reference Dataset returned it
    c = fix["patch_centre"]
    s = [sp for sp in splits if sp[0] <= c[1] - 0 and True][0]
    s0, e0 = [sp for sp in splits if sp[0] - 21 + 128 <= c[1] < sp[1] + 128][0]
    grid = orc.get_data_grid(n_range, int(seabed[s0:e0].max()), s0, e0)
    lo = max(0, grid[0, 1] - 128)
    hi = min(n_pings, grid[-1, 1] + 128)
    d = orc.crop(sv[:, lo:hi].swapaxes(1, 2), (c[0], c[1] - lo), (256, 256), 0)
    d, nf = orc.data_transform(d)
    assert np.abs(d - fix["patch_data"]).max() < 1e-5
    lab = orc.patch_labels(labels[s0:e0].T, {"local": (c[0], c[1] - s0), "global": tuple(c)}, (256, 256),
                           seabed, n_range, int(fix["overlap"]), nf)
    ref = fix["patch_labels"].astype(np.int64)
    same_validity = np.isin(lab, (-100, -70, -50)) == np.isin(ref, (-100, -70, -50))
    assert same_validity.all()
    assert np.array_equal(lab[ref != -30], ref[ref != -30])      # -30 (refine_label_boundary) not modelled


def test_host_planners_match_oracle_and_reference(fix, survey):
    _, _, seabed = survey
    n_pings, n_range = int(fix["n_pings"]), int(fix["n_range"])
    chunks = ti.plan_chunks(0, n_pings, int(fix["preload"]))
    assert np.array_equal(np.array(chunks), fix["splits"])
    s, e = chunks[0]
    assert np.array_equal(ti.plan_grid(n_range, seabed[s:e].max(), s, e, (256, 256), 20), fix["grid0"])
    s, e = chunks[-1]
    assert np.array_equal(ti.plan_grid(n_range, seabed[s:e].max(), s, e, (256, 256), 20), fix["grid_last"])
    # 8192 pings x 1024 range, seabed 900, preload 4096 -> 2 chunks x 95 patches
    assert ti.plan_chunks(0, 8192, 4096) == [(0, 4096), (4096, 8192)]
    g = ti.plan_grid(1024, 900, 0, 4096)
    assert len(g) == 95 and tuple(g[0]) == (107, 107) and tuple(g[1]) == (107, 323) and g[:, 0].max() == 971
    # resume in the middle, chunk sizes stay equal (np.linspace semantics)
    assert ti.plan_chunks(1000, 8192, 4096) == [(1000, 4596), (4596, 8192)]


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32x6", "bf16", "h3p"])
def test_gather_patches_matches_oracle_crop_and_transform(survey, prec):
    from crimac_classifiers_unet_amd import hip
    from crimac_classifiers_unet_amd.hip import call, ptr
    sv, labels, seabed = survey
    lo, hi = 100, 700
    data = torch.from_numpy(np.ascontiguousarray(sv[:, lo:hi])).cuda()
    centres = np.array([[107, 107], [323, 539], [539, 683], [60, 101], [580, 650]], dtype=np.int32)   # some cross borders
    local = centres.copy()
    local[:, 1] -= lo
    loc_d = torch.from_numpy(local).cuda()
    dt = torch.bfloat16 if prec == "bf16" else torch.float32
    out = torch.full((len(centres) * 256 * 256, 16), 3.0, dtype=dt, device="cuda")
    call("crimac_gather_patches", hip.PREC_NAMES[prec], ptr(data), 4, hi - lo, sv.shape[2], ptr(loc_d),
         len(centres), 256, 256, ptr(out), 16)
    torch.cuda.synchronize()
    if prec == "h3p":          # fp16 plane pairs: every 8-channel group holds [8 hi][8 lo], value = hi + lo
        h = out.cpu().view(torch.float16).view(-1, 2, 2, 8).float()
        got = (h[:, :, 0] + h[:, :, 1]).reshape(len(centres), 256, 256, 16).numpy()
    else:
        got = out.float().cpu().numpy().reshape(len(centres), 256, 256, 16)
    for i, c in enumerate(local):
        ref, _ = orc.data_transform(orc.crop(sv[:, lo:hi].swapaxes(1, 2), c, (256, 256), 0))
        tol = 0.3 if prec == "bf16" else (4e-5 if prec == "h3p" else 2e-5)    # bf16: 8 mantissa bits on values up to 75; plane pairs: 22
        assert np.abs(got[i, :, :, :4].transpose(2, 0, 1) - ref).max() < tol
        assert np.abs(got[i, :, :, 4:]).max() == 0


@pytest.mark.gpu
def test_chunk_loop_with_linear_predictor_matches_reference_golden(fix, survey):
    """gather -> (stand-in predictor) -> scatter over every chunk == the reference's out_array."""
    import crimac_classifiers_unet_amd as pkg
    sv, labels, seabed = survey
    reader = FakeZarrReader(sv, labels, seabed)
    n_pings, n_range = reader.shape
    model = pkg.UNet_Baseline(3, 4, precision="f32x6").cuda().eval()
    cp = ti.ChunkPredictor(model, n_range, (256, 256), int(fix["overlap"]), batch_size=4)

    def predict_fn(x, P, H, W):          # x: NHWC [P*H*W,16] dB -> softmax [P,3,H,W] with the stand-in net
        d = x.float().reshape(P, H, W, 16)[..., :4].permute(0, 3, 1, 2).cpu().numpy()
        return torch.from_numpy(np.stack([linear_predictor(di) for di in d])).cuda().contiguous()

    outs = []
    for s, e in ti.plan_chunks(0, n_pings, int(fix["preload"])):
        grid = ti.plan_grid(n_range, seabed[s:e].max(), s, e, (256, 256), int(fix["overlap"]))
        lo, hi = max(0, grid[0, 1] - 128), min(n_pings, grid[-1, 1] + 128)
        cp.load_chunk(reader.get_data_slice(lo, hi - lo), lo, reader.get_label_slice(s, e - s),
                      reader.get_seabed_mask(s, e - s, 0, n_range), s, e)
        outs.append(cp.predict(grid, predict_fn).cpu().numpy())
    full = np.concatenate(outs, axis=2)
    ref = fix["out_f16"].astype(np.float32)
    assert np.array_equal(full != 0, ref != 0)                   # exactly the same pixels written
    assert np.abs(full - ref).max() < 1e-3                       # fixture is float16-rounded


@pytest.mark.gpu
def test_predict_survey_with_unet_matches_oracle(survey):
    """The full path (gather, U-Net f32x6, softmax, scatter) on one chunk vs oracle net + oracle tiling."""
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import synth
    from oracle import unet_oracle as uorc
    sv, labels, seabed = survey
    sv, labels, seabed = sv[:, :440, :300], labels[:440, :300], np.clip(seabed[:440], 0, 230)
    reader = FakeZarrReader(sv, labels, seabed)
    sd = synth.synth_state_dict(seed=0)

    class Pipe:
        frequencies = [18, 38, 120, 200]
        device = torch.device("cuda")
    pipe = Pipe()
    pipe.model = pkg.UNet_Baseline(3, 4, precision="f32x6")
    pipe.model.load_state_dict(sd)
    chunks = list(ti.predict_survey(reader, pipe, (256, 256), 20, 3, 440))
    assert len(chunks) == 1 and chunks[0][:2] == (0, 440)
    got = chunks[0][2]
    torch.set_num_threads(8)

    def net(d):
        return uorc.predict(sd, torch.from_numpy(d)[None], return_softmax=True)[0].numpy()
    ref, grid = orc.predict_chunk(sv, labels, seabed, 0, 440, net)
    assert len(grid) == 6
    assert np.array_equal(got != 0, ref != 0)
    assert np.abs(got - ref).max() < 1e-4


# ---------------------------------------------------------------------------------------------------------------
# memm flavour (save_reader_predictions_memm, save_predict.py:222-265) -- fixture from the reference's own
# DatasetGriddedReader / define_data_transform_test / define_label_transform_test / fill_out_array on a fake Echogram
# (tools/make_golden_tiling_memm.py)
# ---------------------------------------------------------------------------------------------------------------
from tools.fake_reader import FakeEchogram  # noqa: E402


@pytest.fixture(scope="module")
def fix_memm(golden_dir):
    return np.load(os.path.join(golden_dir, "tiling_memm.npz"))


def _memm_case(fix_memm, tag):
    n_pings, n_range, seed = (int(v) for v in fix_memm[tag + "/shape"])
    sv, labels, seabed = synth_survey(n_pings=n_pings, n_range=n_range, seed=seed)
    return np.ascontiguousarray(sv.swapaxes(1, 2)), np.ascontiguousarray(labels.T), seabed


@pytest.mark.parametrize("tag", ["deep", "shallow"])
def test_memm_oracle_matches_reference_golden(fix_memm, tag):
    sv_hw, labels_hw, seabed = _memm_case(fix_memm, tag)
    out = orc.predict_echogram_memm(sv_hw, labels_hw, seabed, linear_predictor)
    assert np.array_equal(out.astype(np.float16), fix_memm[tag + "/out_f16"])
    # one patch as the reference Dataset returned it: border rule (data = 0.0 where the transformed label is -100)
    c = fix_memm[tag + "/patch_centre"]
    d = orc.crop(sv_hw, c, (256, 256), 0)
    d = np.where(np.isfinite(d), d, np.float32(0))
    lab = orc.patch_labels(labels_hw, {"local": tuple(c), "global": tuple(c)}, (256, 256), seabed, sv_hw.shape[1], 20,
                           None, seabed_rule="memm")
    db, _ = orc.data_transform(d)
    db[:, lab == -100] = 0.0
    assert np.abs(db - fix_memm[tag + "/patch_data"]).max() < 1e-5
    ref = fix_memm[tag + "/patch_labels"].astype(np.int64)
    assert np.array_equal(lab[ref != -30], ref[ref != -30])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["deep", "shallow"])
def test_memm_echogram_prediction_matches_reference_golden(fix_memm, tag):
    """crimac_gather_patches_memm (border rule) -> stand-in predictor -> crimac_scatter_patches_ex (Echogram seabed
    rule, seabed vector, float16 output) over a whole echogram == the array the reference would np.save."""
    import crimac_classifiers_unet_amd as pkg
    sv_hw, labels_hw, seabed = _memm_case(fix_memm, tag)
    eg = FakeEchogram(sv_hw, labels_hw, seabed)

    class Pipe:
        frequencies = [18, 38, 120, 200]
        device = torch.device("cuda")
    pipe = Pipe()
    pipe.model = pkg.UNet_Baseline(3, 4, precision="f32x6")
    seen = {}

    def predict_fn(x, P, H, W):
        d = x.float().reshape(P, H, W, 16)[..., :4].permute(0, 3, 1, 2).cpu().numpy()
        seen.setdefault("first", d[0].copy())
        return torch.from_numpy(np.stack([linear_predictor(di) for di in d])).cuda().contiguous()

    out = ti.predict_echogram_memm(eg, pipe, (256, 256), 20, 4, predict_fn=predict_fn)
    ref = fix_memm[tag + "/out_f16"]
    assert out.dtype == np.float64 and out.shape == ref.shape
    assert np.array_equal(out != 0, ref != 0)
    assert np.abs(out - ref.astype(np.float64)).max() < 1e-3            # (the stand-in net runs in fp32 here and there)
    # the border rule, on the first patch of the grid (its top / left rim lies outside the echogram)
    g = fix_memm[tag + "/centres"][0]
    d = orc.crop(sv_hw, g, (256, 256), 0)
    d = np.where(np.isfinite(d), d, np.float32(0))
    lab = orc.patch_labels(labels_hw, {"local": tuple(g), "global": tuple(g)}, (256, 256), seabed, sv_hw.shape[1], 20,
                           None, seabed_rule="memm")
    db, _ = orc.data_transform(d)
    db[:, lab == -100] = 0.0
    assert np.abs(seen["first"] - db).max() < 2e-5
    assert (db[0][lab == -100] == 0).all() and (lab == -100).any()


@pytest.mark.gpu
def test_chunk_loop_with_seabed_vector_and_f16_output_matches_reference_golden(fix, survey):
    """The host-cheap form of the zarr flavour: the seabed VECTOR instead of a [pings, range] mask, float16 output."""
    import crimac_classifiers_unet_amd as pkg
    sv, labels, seabed = survey
    reader = FakeZarrReader(sv, labels, seabed)
    n_pings, n_range = reader.shape
    model = pkg.UNet_Baseline(3, 4, precision="f32x6").cuda().eval()
    cp = ti.ChunkPredictor(model, n_range, (256, 256), int(fix["overlap"]), batch_size=4, out_f16=True)

    def predict_fn(x, P, H, W):
        d = x.float().reshape(P, H, W, 16)[..., :4].permute(0, 3, 1, 2).cpu().numpy()
        return torch.from_numpy(np.stack([linear_predictor(di) for di in d])).cuda().contiguous()

    outs = []
    for s, e in ti.plan_chunks(0, n_pings, int(fix["preload"])):
        grid = ti.plan_grid(n_range, seabed[s:e].max(), s, e, (256, 256), int(fix["overlap"]))
        lo, hi = max(0, grid[0, 1] - 128), min(n_pings, grid[-1, 1] + 128)
        # the seabed vector must cover every ping a patch of the chunk touches: hand over the data slice's range
        cp.load_chunk(reader.get_data_slice(lo, hi - lo), lo, reader.get_label_slice(s, e - s), None, s, e,
                      seabed=reader.get_seabed(lo, hi - lo), seabed_ping0=lo)
        o = cp.predict(grid, predict_fn)
        assert o.dtype == torch.float16
        outs.append(o.cpu().numpy())
    full = np.concatenate(outs, axis=2)
    ref = fix["out_f16"]
    assert np.array_equal(full != 0, ref != 0)                   # exactly the same pixels written
    # the reference's float16 store up to one float16 ulp (the stand-in net sees log10f of the GPU, not numpy's)
    assert np.abs(full.astype(np.float32) - ref.astype(np.float32)).max() <= 2.0 ** -10


def _holey_case():
    sv, labels, seabed = synth_survey()
    sv, labels, seabed = sv[:, :500], labels[:500], seabed[:500]
    mask = holey_seabed_mask(seabed, sv.shape[2])
    return sv, labels, FakeZarrReader(sv, labels, seabed, mask=mask), mask


def test_oracle_with_the_readers_2d_seabed_mask_matches_reference_golden():
    """Pings without a detected bottom / masks with holes: the reference reads the 2-D mask (mask_label_seabed.py:47-49);
    fixture from the reference's own Dataset + transforms + fill_out_array (tools/make_golden_tiling_mask.py)."""
    fix = np.load(os.path.join(ROOT, "tests", "golden", "tiling_mask.npz"))
    sv, labels, reader, mask = _holey_case()
    out, _ = orc.predict_chunk(sv, labels, reader.seabed, 0, 500, linear_predictor, (256, 256), 20, seabed_mask=mask)
    ref = fix["out_f16"].astype(np.float32)
    assert np.array_equal(out[0] != 0, ref[0] != 0) and np.abs(out - ref).max() < 1e-3
    vec, _ = orc.predict_chunk(sv, labels, reader.seabed, 0, 500, linear_predictor, (256, 256), 20)
    assert ((vec[0] != 0) != (ref[0] != 0)).sum() > 1000          # the seabed-vector rule is NOT the reference here


def test_seabed_vector_is_only_used_where_it_equals_the_mask():
    n_range = 64
    seabed = np.array([10, 20, 30, 40, 50, 60], dtype=np.int32)
    full = (np.arange(n_range)[None, :] >= seabed[:, None]).astype(np.uint8)

    class R:
        def __init__(self, m):
            self.m = m

        def get_seabed_mask(self, idx_ping, n_pings, idx_range=None, n_range=None, return_numpy=False, seabed_pad=0):
            return self.m[idx_ping:idx_ping + n_pings].astype(np.float64)

    # the mask is the threshold: vector path, unchanged
    sb, mask = ti.seabed_vector_or_mask(R(full), 1, 5, n_range, seabed.copy(), 0)
    assert mask is None and np.array_equal(sb, seabed)
    # a ping without a detected bottom (all-zero column, argmax 0): nothing of it is below the seabed
    m = full.copy(); m[2] = 0
    vec = m.argmax(axis=1).astype(np.int32)
    sb, mask = ti.seabed_vector_or_mask(R(m), 1, 5, n_range, vec.copy(), 0)
    assert mask is None and sb[2] == n_range and np.array_equal(np.delete(sb, 2), np.delete(vec, 2))
    # ... outside the pings the chunk writes it is left alone
    sb, mask = ti.seabed_vector_or_mask(R(m), 3, 5, n_range, vec.copy(), 0)
    assert mask is None and np.array_equal(sb, vec)
    # a hole: not a threshold -> the mask itself
    m = full.copy(); m[3, 45:50] = 0
    sb, mask = ti.seabed_vector_or_mask(R(m), 1, 5, n_range, m.argmax(axis=1).astype(np.int32), 0)
    assert mask is not None and mask.dtype == np.uint8 and np.array_equal(mask, m[1:5])
    # a reader without the member keeps the vector
    assert ti.seabed_vector_or_mask(object(), 0, 6, n_range, seabed, 0)[1] is None


@pytest.mark.gpu
@pytest.mark.parametrize("preload", [500, 250])
def test_predict_survey_follows_the_readers_seabed_mask(preload):
    """predict_survey on a reader whose stored mask has no-bottom pings and holes == the reference golden (chunks of 250
    pings: the first has only no-bottom pings -> seabed vector with n_range; the second has holes -> mask upload)."""
    import types
    import crimac_classifiers_unet_amd as pkg
    fix = np.load(os.path.join(ROOT, "tests", "golden", "tiling_mask.npz"))
    sv, labels, reader, mask = _holey_case()
    model = pkg.UNet_Baseline(3, 4, precision="f32x6").cuda().eval()
    pipe = types.SimpleNamespace(model=model, device=torch.device("cuda"), frequencies=[18, 38, 120, 200])

    def predict_fn(x, P, H, W):
        d = x.float().reshape(P, H, W, 16)[..., :4].permute(0, 3, 1, 2).cpu().numpy()
        return torch.from_numpy(np.stack([linear_predictor(di) for di in d])).cuda().contiguous()

    ref = fix["out_f16"].astype(np.float32)
    if preload == 500:
        chunks = list(ti.predict_survey(reader, pipe, (256, 256), 20, 4, preload, predict_fn=predict_fn))
        full = np.concatenate([c[2] for c in chunks], axis=2)
        assert np.array_equal(full != 0, ref != 0) and np.abs(full - ref).max() < 1e-3
    else:
        # chunked: the grid (hence which patch writes a pixel) differs from the one-chunk golden, the written SET and
        # the seabed rule do not -- compare with the oracle run chunk by chunk on the reader's mask
        for s, e, out in ti.predict_survey(reader, pipe, (256, 256), 20, 4, preload, predict_fn=predict_fn):
            o, _ = orc.predict_chunk(sv, labels, reader.seabed, s, e, linear_predictor, (256, 256), 20, seabed_mask=mask)
            assert np.array_equal(out != 0, o != 0) and np.abs(out - o).max() < 1e-3


@pytest.mark.gpu
def test_configs3_chunk_of_4096_pings_covers_the_water_column():
    """BASELINE configs[3] geometry at size: one chunk of 4096 pings x 1024 range, flat seabed 900
    -> 95 patches; predict_survey (reader thread, pinned staging, copy stream, GPU mask from the seabed vector,
    float16 result) writes 100 % of [0:910) x chunk and nothing below."""
    import types
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import synth
    reader = synth.SyntheticSurveyReader(n_pings=8192, n_range=1024, seabed_index=900, block=4096)
    model = pkg.UNet_Baseline(3, 4, precision="bf16")
    model.load_state_dict(synth.synth_state_dict(seed=0))
    pipe = types.SimpleNamespace(model=model, device=torch.device("cuda"), frequencies=[18, 38, 120, 200])
    chunks = list(ti.predict_survey(reader, pipe, (256, 256), 20, 32, 4096, out_dtype=np.float16))
    assert [c[:2] for c in chunks] == [(0, 4096), (4096, 8192)]
    assert len(ti.plan_grid(1024, 900, 0, 4096)) == 95
    for s, e, out in chunks:
        assert out.dtype == np.float16 and out.shape == (2, 1024, 4096)
        assert (out[0, :910] != 0).all() and (out[1, :910] != 0).all()          # softmax > 0: written everywhere
        assert (out[:, 910:] == 0).all()
        assert np.isfinite(out).all() and out.max() <= 1.0
    # the two chunks see the same tiled block of data: same predictions in their interiors (patch grids are shifted
    # by the chunk origin, so compare a column range both grids cover identically)
    s32 = list(ti.predict_survey(reader, pipe, (256, 256), 20, 32, 4096, out_dtype=np.float32))
    assert np.abs(s32[0][2].astype(np.float16).astype(np.float32) - chunks[0][2].astype(np.float32)).max() <= 1e-3


def _tiled_rank_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", CRIMAC_DIST_BACKEND="gloo")
    import types
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import parallel, synth
    parallel.init_distributed(backend="gloo")          # two ranks share the one GPU of the box: gloo, not RCCL
    sv, labels, seabed = synth_survey()
    sv, labels, seabed = sv[:, :700, :300], labels[:700, :300], np.clip(seabed[:700], 0, 230)
    reader = FakeZarrReader(sv, labels, seabed)
    model = pkg.UNet_Baseline(3, 4, precision="f32x6")
    model.load_state_dict(synth.synth_state_dict(seed=0))
    pipe = types.SimpleNamespace(model=model, device=torch.device("cuda"), frequencies=[18, 38, 120, 200])
    chunks = list(ti.predict_survey(reader, pipe, (256, 256), 20, 2, 350, out_dtype=np.float16, shard="patch"))
    if rank == 0:
        out["chunks"] = [(s, e, o.copy()) for s, e, o in chunks]
    # chunk sharding: rank r owns chunks r, r + N, ...; no collective; together the ranks cover the survey once
    mine = list(ti.predict_survey(reader, pipe, (256, 256), 20, 2, 350, out_dtype=np.float16))      # (default: own chunks)
    out[f"own{rank}"] = [(s, e, o.copy()) for s, e, o in mine]
    # opt-in (ordered_to_rank0=True): chunk-sharded compute, ordered hand-off -- rank 0 yields the WHOLE survey in
    # ping order (what a sequential append_to_zarr writer on rank 0 needs), the other ranks yield nothing
    ordered = list(ti.predict_survey(reader, pipe, (256, 256), 20, 2, 175, out_dtype=np.float16, ordered_to_rank0=True))
    out[f"ordered{rank}"] = [(s, e, o.copy()) for s, e, o in ordered]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_share_the_patches_of_a_chunk_and_merge_exactly():
    """SURVEY.md §8e inference partition, both forms: patch p of a chunk -> rank p % 2, the per-rank float16 outputs
    summed (disjoint interiors); and chunk c -> rank c % 2 with no collective at all.  Both bit-identical to the
    single-process result."""
    import socket
    import types
    import torch.multiprocessing as mp
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import synth
    sv, labels, seabed = synth_survey()
    sv, labels, seabed = sv[:, :700, :300], labels[:700, :300], np.clip(seabed[:700], 0, 230)
    reader = FakeZarrReader(sv, labels, seabed)
    model = pkg.UNet_Baseline(3, 4, precision="f32x6")
    model.load_state_dict(synth.synth_state_dict(seed=0))
    pipe = types.SimpleNamespace(model=model, device=torch.device("cuda"), frequencies=[18, 38, 120, 200])
    single = list(ti.predict_survey(reader, pipe, (256, 256), 20, 2, 350, out_dtype=np.float16))
    assert len(single) == 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_tiled_rank_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        multi = out["chunks"]
        own = [out["own0"], out["own1"]]
        ordered = [out["ordered0"], out["ordered1"]]
    for (s0, e0, o0), (s1, e1, o1) in zip(single, multi):
        assert (s0, e0) == (s1, e1) and np.array_equal(o0, o1) and (o0 != 0).any()
    # chunk sharding: rank 0 produced chunk 0, rank 1 chunk 1, each bit-identical to the single-process result
    assert [c[:2] for c in own[0]] == [single[0][:2]] and [c[:2] for c in own[1]] == [single[1][:2]]
    assert np.array_equal(own[0][0][2], single[0][2]) and np.array_equal(own[1][0][2], single[1][2])
    # ordered hand-off (4 chunks over 2 ranks): rank 0 got all four, in ping order, bit-identical; rank 1 nothing
    single4 = list(ti.predict_survey(reader, pipe, (256, 256), 20, 2, 175, out_dtype=np.float16))
    assert len(single4) == 4 and ordered[1] == []
    assert [c[:2] for c in ordered[0]] == [c[:2] for c in single4]
    for (_, _, a), (_, _, b) in zip(ordered[0], single4):
        assert a.dtype == np.float16 and np.array_equal(a, b)


# ---- metadata planes (late metadata injection) --------------------------------------------------------------
@pytest.fixture(scope="module")
def fix_meta():
    return np.load(os.path.join(ROOT, "tests", "golden", "meta_planes.npz"))


def _meta_channels(name):
    mc = {k: True for k in orc.META_KEYS}
    if name == "subset":
        mc.update(portion_day=False, depth_rel=False)
    return mc


@pytest.mark.parametrize("name", ["all", "subset"])
def test_metadata_planes_oracle_matches_reference_golden(fix_meta, name):
    """oracle/tiling_oracle.meta_planes == the reference's get_crop_memmap (tools/make_golden_meta.py: equal bit for bit
    in float64 there; the fixture stores the float32 the batch is cast to)."""
    for i, c in enumerate(fix_meta["centres"]):
        got = orc.meta_planes(c, (256, 256), _meta_channels(name), float(fix_meta["portion_year"]), fix_meta["portion_day"],
                              fix_meta["time_diff"], fix_meta["seabed"])
        assert np.array_equal(got.astype(np.float32), fix_meta["planes_" + name][i], equal_nan=True)
    got = orc.meta_planes(fix_meta["shallow_centre"], (256, 256), _meta_channels("all"), float(fix_meta["portion_year"]),
                          fix_meta["portion_day"], fix_meta["time_diff"], fix_meta["shallow_seabed"])
    assert np.array_equal(got.astype(np.float32), fix_meta["shallow_planes"], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["all", "subset"])
def test_metadata_planes_kernel_matches_reference_golden(fix_meta, name):
    """crimac_meta_planes (float64 arithmetic on the GPU, one rounding) == the reference's planes cast to float32: exact for
    every plane but sin / cos of the time of day, where the device libm may differ from numpy's in the last bit."""
    src = ti.MetaSource(_meta_channels(name), float(fix_meta["portion_year"]), fix_meta["portion_day"], fix_meta["time_diff"],
                        fix_meta["seabed"], "cuda")
    cen = torch.as_tensor(fix_meta["centres"].astype(np.int32)).cuda()
    got = src.planes(cen, (256, 256)).cpu().numpy()
    ref = fix_meta["planes_" + name]
    assert got.shape == ref.shape
    trig = [1, 2] if name == "all" else []
    for c in range(ref.shape[1]):
        if c in trig:
            assert np.abs(got[:, c] - ref[:, c]).max() <= 1.2e-7
        else:
            assert np.array_equal(got[:, c], ref[:, c], equal_nan=True), c
    with pytest.raises(ValueError):
        ti.MetaSource({"portion_year": True}, 0.5, fix_meta["portion_day"], fix_meta["time_diff"], fix_meta["seabed"], "cuda")


@pytest.mark.gpu
def test_late_metadata_injection_through_tiled_inference_builds_the_planes_on_the_gpu():
    """predict_echogram_memm with a UNet_LateMetInject model: the metadata planes of every crop come from
    crimac_meta_planes.  Equal to feeding the SAME network the oracle's planes (reference semantics, golden above) crop by
    crop and scattering with the oracle's fill_out_array."""
    import types
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import synth
    sv, labels, seabed = synth_survey(n_pings=520, n_range=300, seed=21)
    sv_hw, labels_hw = np.ascontiguousarray(sv.swapaxes(1, 2)), np.ascontiguousarray(labels.T)
    seabed = np.clip(seabed, 40, 280)
    eg = FakeEchogram(sv_hw, labels_hw, seabed)
    rng = np.random.Generator(np.random.PCG64(9))
    tv = 737000.5 + np.cumsum(rng.uniform(5e-6, 9e-6, size=520))
    eg.portion_of_day_vector = tv % 1
    eg.portion_of_year_scalar = 0.61
    eg.time_vector_diff = np.concatenate((np.diff(tv), [tv[-1] - tv[-2]])) / 6e-6 - 1
    mc = {k: True for k in orc.META_KEYS}
    model = pkg.UNet_LateMetInject(3, 4, 7, precision="f32x6")
    model.load_state_dict(synth.synth_state_dict(seed=3, meta_in_channels=7))
    model.cuda().eval()
    pipe = types.SimpleNamespace(model=model, device=torch.device("cuda"), frequencies=[18, 38, 120, 200])
    out = ti.predict_echogram_memm(eg, pipe, (256, 256), 20, 4, meta_channels=mc)

    def net(db, centre):
        meta = orc.meta_planes(centre, (256, 256), mc, eg.portion_of_year_scalar, eg.portion_of_day_vector,
                               eg.time_vector_diff, eg._seabed).astype(np.float32)
        with torch.no_grad():
            z = model(torch.from_numpy(db[None]).cuda(), torch.from_numpy(meta[None]).cuda())
            return torch.softmax(z, 1)[0].cpu().numpy()

    n_range, n_pings = eg.shape
    grid = orc.get_data_grid(n_range, int(seabed.max()), 0, n_pings, (256, 256), 20)
    ref = np.zeros([2, n_range, n_pings])
    for c in grid:
        c = np.array(c)
        d = orc.crop(sv_hw, c, (256, 256), 0)
        d = np.where(np.isfinite(d), d, d.dtype.type(0))
        lab = orc.patch_labels(labels_hw, {"local": tuple(c), "global": tuple(c)}, (256, 256), seabed, n_range, 20, None,
                               seabed_rule="memm")
        db, _ = orc.data_transform(d)
        db[:, lab == orc.LABEL_BOUNDARY_VAL] = 0.0
        orc.fill_out_array(ref, net(db.astype(np.float32), c).astype(np.float16), lab, c, 0)
    assert np.array_equal(out != 0, ref != 0) and (out != 0).any()
    assert np.abs(out - ref).max() <= 2e-3                      # float16-rounded probabilities
