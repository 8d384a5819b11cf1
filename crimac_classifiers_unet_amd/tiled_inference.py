"""Tiled whole-survey inference on the GPU: the ``save_predict.py`` path of the reference.

Mirrors ``save_survey_predictions_zarr`` (crimac_unet/pipeline_train_predict/save_predict.py:137-220, the zarr
preload flavour) and ``save_reader_predictions_memm`` (:222-265, the memmap flavour): the survey is cut into
chunks of at most ``preload_n_pings`` pings (utils/preload_data_split.py:22-30), every chunk is gridded into
overlapping patches (batch/samplers/gridded.py:22-54), each patch is cropped + dB-transformed, pushed through the
U-Net + softmax, and the valid interior of its SANDEEL / OTHER probabilities is scattered into a
``[2, range, pings]`` array (save_predict.py:41-65).

Here a chunk is uploaded ONCE and stays in HBM; crop, transform, forward, softmax and scatter all run on the GPU
(``crimac_gather_patches`` -> U-Net engine -> ``crimac_scatter_patches_ex``); only the finished
``[2, range, pings]`` chunk comes back -- as float16, which is what the reference stores (save_predict.py:212,
:252).  The reference does the crop / transform / ``argwhere`` scatter per patch in numpy DataLoader workers and
moves 786 kB of softmax per patch over PCIe.

Host side of ``predict_survey`` (what bounded configs[3] in round 1): the reader thread copies the next chunk
straight into PINNED staging buffers, the upload runs on a copy stream, the seabed mask is evaluated in the scatter
kernel from the seabed vector (no [pings, range] mask is built or uploaded), and the result returns through a
pinned float16 buffer.

With ``torch.distributed`` initialised, patches of a chunk are dealt round-robin to the ranks
(``parallel.shard_indices``) and the per-rank float16 outputs are summed (valid interiors are disjoint, x + 0 is
exact).  Writing zarr / npy is the caller's business: ``predict_survey`` yields numpy chunks
that a caller appends with the reference's ``create_xarray_ds_predictions`` / ``append_to_zarr``;
``predict_echogram_memm`` returns the array ``save_reader_predictions_memm`` would ``np.save``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip, parallel
from .hip import call, ptr

SEABED_PAD = 10        # mask_label_seabed.py:50-52
SEABED_MARGIN = 50     # gridded.py:150-156
INTERNAL_BATCH = 96    # patches per forward call (eval mode: results do not depend on the batch size)


def plan_chunks(start_ping, n_pings, preload_n_pings):
    """``get_data_split([[start, n_pings]], preload_n_pings)`` (preload_data_split.py:22-30)."""
    max_n = preload_n_pings if preload_n_pings > 0 else 5000          # save_predict.py:160-166
    n_splits = int(np.ceil((n_pings - start_ping) / max_n))
    edges = np.linspace(start_ping, n_pings, n_splits + 1).astype(int)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(n_splits)]


def plan_grid(n_range, max_seabed, start_ping, end_ping, patch_size=(256, 256), patch_overlap=20):
    """Patch centres (range idx, ping idx), ping fastest: ``get_data_grid(mode='all')``
    (gridded.py:35-54) with the range extent capped at max seabed + 50 (:150-159)."""
    end_range = min(n_range, int(max_seabed) + SEABED_MARGIN)
    assert end_range > 0 and end_ping > start_ping
    pw, ph = patch_size
    ys = np.arange(-(patch_overlap + 1), end_range - (patch_overlap + 1), ph - 2 * patch_overlap) + ph // 2
    xs = np.arange(start_ping - (patch_overlap + 1), end_ping - (patch_overlap + 1),
                   pw - 2 * patch_overlap) + pw // 2
    return np.array(np.meshgrid(ys, xs)).T.reshape(-1, 2)


META_FLAGS = {"portion_year": 1, "portion_day": 2, "time_diff": 4, "depth_rel": 8, "depth_abs_surface": 16,
              "depth_abs_seabed": 32}          # planes come out in this order (batch/dataset.py:288-351)


class MetaSource:
    """The per-ping vectors the metadata planes of an echogram are built from (data_reader.py:98-100), resident on the
    GPU: ``crimac_meta_planes`` turns them into the ``[P, Cm, H, W]`` planes of a batch of crops -- what the reference's
    ``get_crop_memmap`` builds per patch in numpy DataLoader workers (batch/dataset.py:288-351)."""

    def __init__(self, meta_channels, portion_year, portion_day_vector, time_vector_diff, seabed, device):
        if set(meta_channels) != set(META_FLAGS) or not all(isinstance(v, bool) for v in meta_channels.values()):
            raise ValueError(f"meta_channels must be a dict of booleans with the keys {sorted(META_FLAGS)}")    # dataset.py:60-66
        self.flags = sum(f for k, f in META_FLAGS.items() if meta_channels[k])
        if self.flags == 0:
            raise ValueError("no metadata channel is switched on")
        self.n_planes = sum((2 if k == "portion_day" else 1) for k in META_FLAGS if meta_channels[k])
        self.portion_year = float(portion_year)

        def dev(a, dt):
            return torch.as_tensor(np.ascontiguousarray(np.asarray(a)).astype(dt)).to(device)
        self.portion_day = dev(portion_day_vector, np.float64)
        self.time_diff = dev(time_vector_diff, np.float64)
        self.seabed = dev(seabed, np.int64)

    @classmethod
    def from_echogram(cls, echogram, meta_channels, device):
        return cls(meta_channels, echogram.portion_of_year_scalar, echogram.portion_of_day_vector,
                   echogram.time_vector_diff, echogram._seabed, device)

    def planes(self, centres_dev, patch_size):
        """centres_dev int32 [P, 2] (range idx, ping idx) on the GPU -> float32 [P, Cm, H, W]."""
        P = centres_dev.shape[0]
        H, W = int(patch_size[1]), int(patch_size[0])
        out = torch.empty((P, self.n_planes, H, W), dtype=torch.float32, device=centres_dev.device)
        with torch.cuda.device(centres_dev.device):
            call("crimac_meta_planes", ptr(centres_dev), P, H, W, self.flags, self.portion_year, ptr(self.portion_day),
                 self.portion_day.numel(), ptr(self.time_diff), self.time_diff.numel(), ptr(self.seabed),
                 self.seabed.numel(), ptr(out))
        return out


class ChunkPredictor:
    """GPU state of one preloaded chunk: data, labels, seabed; gathers, predicts, scatters."""

    def __init__(self, model, n_range, patch_size=(256, 256), patch_overlap=20, batch_size=32, out_f16=False):
        self.model = model
        self.engine = model.infer_engine        # (eval-mode forwards: the model's inference precision)
        self.n_range = n_range
        self.patch_size = tuple(int(v) for v in patch_size)
        self.patch_overlap = int(patch_overlap)
        self.batch_size = int(batch_size)
        self.out_f16 = bool(out_f16)
        self.flavour = "zarr"
        self.seabed = self.mask = None
        self.meta_source = None          # MetaSource: needed by a UNet_LateMetInject model (metadata planes per crop)

    def _device(self):
        return self.engine.device or next(self.model.parameters()).device

    def load_chunk(self, data, data_ping0, labels, seabed_mask, start_ping, end_ping, seabed=None, seabed_ping0=None,
                   flavour="zarr", stream=None):
        """data [C, pings, range] fp32 (global ping of column 0 = data_ping0; numpy or a pinned CPU tensor);
        labels [end-start, range] (or None); for pings [start_ping, end_ping) either ``seabed_mask``
        [end-start, range] uint8/bool (1 below the seabed, as the reader's get_seabed_mask(..., seabed_pad=0)) or
        -- cheaper, nothing to build on the host -- ``seabed`` [n] int seabed index per ping starting at global ping
        ``seabed_ping0`` (default start_ping).  ``stream``: copy stream for the uploads (the caller orders it)."""
        dev = self._device()
        nb = stream is not None

        def up(a, dtype):
            if a is None:
                return None
            if torch.is_tensor(a):          # already a tensor (pinned host staging or device resident): keep its dtype
                t = a
            else:
                t = torch.as_tensor(np.ascontiguousarray(np.asarray(a).astype(dtype, copy=False)))
            if stream is not None:
                with torch.cuda.stream(stream):
                    return t.to(dev, non_blocking=nb)
            return t.to(dev)

        self.data = up(data, np.float32)
        self.data_ping0 = int(data_ping0)
        self.labels = up(labels, np.int16)
        self.mask = up(seabed_mask, np.uint8)
        self.seabed = up(seabed, np.int32)
        self.seabed_ping0 = int(start_ping if seabed_ping0 is None else seabed_ping0)
        self.start_ping, self.end_ping = int(start_ping), int(end_ping)
        self.flavour = flavour
        self.out = torch.zeros((2, self.n_range, self.end_ping - self.start_ping),
                               dtype=torch.float16 if self.out_f16 else torch.float32, device=dev)

    def predict(self, grid, predict_fn=None, centres_dev=None, share_patches=True):
        """Run all patches of ``grid`` ([P,2] global centres) and scatter them into ``self.out``.

        ``share_patches`` (with torch.distributed initialised): the patches of THIS chunk are dealt round-robin to the
        ranks and the per-rank outputs are summed (one collective per chunk); False: this rank computes the whole chunk
        on its own (``predict_survey`` with chunk sharding: the ranks own different chunks, no collective at all).

        ``predict_fn(x_nhwc, P, H, W) -> probs [P,3,H,W]`` overrides the network (tests).
        ``centres_dev``: int32 [2, P, 2] on the GPU = (global centres, centres relative to the data slice), uploaded
        by the caller (a pageable host-to-device copy here would block the host until the stream has drained and
        serialise the enqueue of a chunk with the execution of the previous one)."""
        eng = self.engine
        eng.bind()
        ph, pw = self.patch_size[1], self.patch_size[0]
        C = self.data.shape[0]
        world, rank = 1, 0
        if share_patches and torch.distributed.is_available() and torch.distributed.is_initialized():
            world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        mine = parallel.shard_indices(len(grid), rank, world)
        memm = self.flavour == "memm"
        step = max(self.batch_size, INTERNAL_BATCH) if predict_fn is None else self.batch_size
        for b0 in range(0, len(mine), step):
            idx = mine[b0:b0 + step]
            P = len(idx)
            if centres_dev is not None and world == 1:
                cen_d, loc_d = centres_dev[0, b0:b0 + P], centres_dev[1, b0:b0 + P]
            else:
                cen = np.asarray(grid)[idx].astype(np.int32)
                local = cen.copy()
                local[:, 1] -= self.data_ping0
                both = torch.from_numpy(np.ascontiguousarray(np.stack([cen, local]))).to(self.data.device)
                cen_d, loc_d = both[0], both[1]
            x = eng._buf("tiled.x", (P * ph * pw, 16))
            if memm:
                call("crimac_gather_patches_memm", eng.prec, ptr(self.data), C, self.data.shape[1], self.n_range,
                     ptr(loc_d), P, ph, pw, ptr(x), 16, ptr(self.labels))
            else:
                call("crimac_gather_patches", eng.prec, ptr(self.data), C, self.data.shape[1], self.n_range,
                     ptr(loc_d), P, ph, pw, ptr(x), 16)
            if predict_fn is not None:
                probs = predict_fn(x, P, ph, pw)
            elif eng.lmi:
                if self.meta_source is None:
                    raise ValueError("a UNet_LateMetInject model needs the metadata planes: set ChunkPredictor.meta_source "
                                     "(MetaSource.from_echogram(...))")
                meta = self.meta_source.planes(cen_d.contiguous(), self.patch_size)
                probs = eng.forward_nhwc(x, P, ph, pw, False, softmax=True, meta=meta)
            else:
                probs = eng.forward_nhwc_eval_split(x, P, ph, pw, softmax=True)
            call("crimac_scatter_patches_ex", ptr(probs), probs.shape[1], ptr(cen_d), P, ph, pw,
                 self.patch_overlap, self.start_ping, self.end_ping - self.start_ping, self.n_range,
                 ptr(self.labels), ptr(self.mask), self.start_ping, self.end_ping - self.start_ping,
                 ptr(self.seabed), self.seabed_ping0, 0 if self.seabed is None else self.seabed.numel(),
                 None if memm else ptr(self.data[0]), self.data_ping0, self.data.shape[1], SEABED_PAD,
                 1 if memm else 0, ptr(self.out), 1 if self.out_f16 else 0)
        if world > 1:
            torch.distributed.all_reduce(self.out)      # interiors are disjoint: sum == union (exact in fp16 too)
        return self.out


def seabed_vector_or_mask(reader, s, e, n_range, sb, sb_ping0):
    """The scatter kernel rebuilds the reader's 2-D seabed mask (``get_seabed_mask``: 1 below the seabed, what
    ``mask_label_seabed`` uses, mask_label_seabed.py:47-49) from the seabed VECTOR as ``range >= seabed[ping]``.  The zarr
    reader derives that vector as ``argmax(range)`` of the stored mask (data_reader.py:864-865), so the two agree only
    where the mask of a ping is exactly "zeros, then ones to the end": a ping with NO detected bottom has an all-zero
    mask (``fillna(0)``) and argmax 0 -- the vector rule would mask its whole water column, the reference masks
    nothing -- a mask with holes is not a threshold at all, and a reader whose ``get_seabed`` is independent of its mask
    (the memmap reader's seabed.npy) may simply disagree with it.  Checked here per chunk, EXACTLY, on the reader's own
    mask (read once) for the pings [s, e) the chunk writes: the mask must equal ``arange(n_range) >= vector`` element for
    element; no-bottom pings get seabed = n_range (nothing below it); anything the vector cannot express is returned as
    uint8 [e - s, n_range] to be uploaded instead (``ChunkPredictor.load_chunk``).

    Returns (seabed vector to upload, mask or None).  ``sb`` covers pings [sb_ping0, sb_ping0 + len(sb))."""
    if not hasattr(reader, "get_seabed_mask"):
        return sb, None
    m = reader.get_seabed_mask(int(s), int(e - s), 0, int(n_range), return_numpy=True)
    m = np.asarray(getattr(m, "values", m))
    if m.shape != (e - s, n_range):
        raise ValueError(f"get_seabed_mask returned {m.shape}, expected {(e - s, n_range)}")
    below = m != 0
    vec = np.asarray(sb[s - sb_ping0:e - sb_ping0]).astype(np.int64)
    none = ~below.any(axis=1)
    vec_eff = np.where(none, n_range, vec)                   # no bottom detected: nothing lies below it
    if np.array_equal(below, np.arange(n_range)[None, :] >= vec_eff[:, None]):
        if none.any():
            sb = sb.copy()
            sb[s - sb_ping0:e - sb_ping0] = vec_eff
        return sb, None
    return sb, np.ascontiguousarray(below.astype(np.uint8))


class _OrderedHandoff:
    """Chunk-sharded inference, one writer: ranks r > 0 send each finished chunk to rank 0, which yields the whole survey
    in ping order (``predict_survey(ordered_to_rank0=True)``).  Point-to-point only: ``isend`` / ``irecv`` of one
    ``[2, n_range, pings]`` tensor per chunk -- under RCCL ("nccl") the DEVICE tensor the scatter kernel wrote (device to
    device over xGMI; the sender does no D2H copy at all), under gloo a host tensor.  Chunk i belongs to rank i % N, so
    rank 0 walks the survey in rounds of N chunks: it posts the N - 1 receives of a round, runs its own chunk of the
    round (its pipeline already works on the next one), then takes the received chunks in rank order."""

    MAX_IN_FLIGHT = 2          # sends a rank keeps outstanding before it waits for the oldest (back-pressure)

    def __init__(self, rank, world, all_chunks, n_range, out_dtype):
        self.rank, self.world, self.all_chunks, self.n_range = rank, world, all_chunks, n_range
        self.device_tensors = torch.distributed.get_backend() == "nccl"
        self.tdtype = torch.float16 if out_dtype == np.float16 else torch.float32
        self.inflight = []
        self.bufs = {}

    # -- ranks r > 0 --------------------------------------------------------------------------------------------
    def send(self, out):
        t = out if self.device_tensors else out.cpu()       # (cp.out is a fresh tensor per chunk: safe to keep)
        self.inflight.append((torch.distributed.isend(t.contiguous(), dst=0), t))
        while len(self.inflight) > self.MAX_IN_FLIGHT:
            self.inflight.pop(0)[0].wait()

    def flush(self):
        for req, _ in self.inflight:
            req.wait()
        self.inflight = []
        if self.device_tensors:
            torch.cuda.current_stream().synchronize()

    # -- rank 0 -------------------------------------------------------------------------------------------------
    def _buf(self, r, n_pings, device):
        key = (r, n_pings)
        if key not in self.bufs:
            self.bufs[key] = torch.empty((2, self.n_range, n_pings), dtype=self.tdtype,
                                         device=device if self.device_tensors else "cpu")
        return self.bufs[key]

    def merge(self, own):
        """``own``: rank 0's generator over ITS chunks (0, N, 2N, ...) -> every chunk of the survey in ping order."""
        device = torch.device("cuda", torch.cuda.current_device()) if self.device_tensors else None
        own = iter(own)
        for k0 in range(0, len(self.all_chunks), self.world):
            reqs = []
            for r in range(1, self.world):
                if k0 + r < len(self.all_chunks):
                    s, e = self.all_chunks[k0 + r]
                    buf = self._buf(r, e - s, device)
                    reqs.append((s, e, buf, torch.distributed.irecv(buf, src=r)))
            yield next(own)
            for s, e, buf, req in reqs:
                req.wait()
                yield s, e, buf.cpu().numpy() if self.device_tensors else buf.numpy().copy()      # (.cpu() is a fresh array; the host buffer is reused)


_STAGING = {}          # (device, sizes) -> pinned / device staging buffers of predict_survey


def release_staging():
    """Give back the staging buffers ``predict_survey`` keeps between surveys of the same geometry (for a 4096-ping
    preload: ~0.5 GB of pinned host memory and ~0.3 GB of device memory)."""
    _STAGING.clear()


def predict_survey(reader, segpipe, patch_size, patch_overlap, batch_size, preload_n_pings,
                   start_ping=0, labels_available=True, out_dtype=np.float32, stats=None, predict_fn=None,
                   shard="chunk", ordered_to_rank0=False, **kwargs):
    """Generator over chunks: yields ``(start_ping, end_ping, out[2, n_range, end-start] numpy)``.

    Multi-GPU (torch.distributed initialised, one process per GPU).  EVERY rank must iterate the
    generator to its end.
      ``shard="chunk"`` (default) -- rank r owns chunks r, r + N, r + 2N, ...: every rank reads, uploads and predicts
          only ITS ping ranges; the reader I/O and the PCIe traffic scale with the ranks too.  What the generators yield
          is set by ``ordered_to_rank0``:
            True (opt-in) -- the finished float16 / float32 chunks are handed to rank 0
              point-to-point (one send per chunk, 16.8 MB as float16; RCCL: device to device over xGMI, off the compute
              stream; gloo: host tensors) and RANK 0 YIELDS EVERY CHUNK OF THE SURVEY IN PING ORDER, the other ranks
              yield nothing: the reference's strictly sequential writer (``append_to_zarr`` with
              ``append_dim='ping_time'`` and resume by ``sizes['ping_time']``, save_predict.py:107-134) runs unchanged on
              rank 0, and a caller that writes on rank 0 only loses nothing.  Rank 0 MUST then consume every chunk: the
              other ranks block in their sends (gloo) or queue them on RCCL's stream until rank 0 has posted the matching
              receive -- a consumer that stops early on rank 0 (exception, ``break``, resume logic) strands them until the
              process group's timeout, which is why this form has to be asked for;
            False (default) -- no communication at all: each rank yields its OWN chunks only (disjoint ping ranges); for
              callers that write regions themselves.
      ``shard="patch"`` -- every rank walks every chunk, takes patches p = rank (mod N) of it and the per-rank float16
          outputs are summed (one all-reduce of the chunk per chunk): every rank yields every chunk.

    ``reader``: the reference's zarr reader API (shape, get_data_slice, get_label_slice, get_seabed);
    ``segpipe``: a ``SegPipeUNet`` with loaded parameters.  ``out_dtype=np.float16`` returns what the reference
    stores (save_predict.py:212) and halves the bytes that come back; float32 (default) keeps full probabilities.
    ``predict_fn(x_nhwc, P, H, W) -> probs [P,3,H,W]`` replaces the network (tests of the plumbing).

    Three stages overlap: a host thread reads chunk i+1 from ``reader`` straight into pinned staging buffers
    (numpy / zarr I/O, no GPU calls) while the GPU uploads (copy stream), gathers, predicts and scatters chunk i, and
    the result of chunk i-1 comes back through a pinned buffer (asynchronous D2H) before it is handed to the caller.
    """
    from concurrent.futures import ThreadPoolExecutor
    n_pings, n_range = reader.shape
    dev = segpipe.device
    model = segpipe.model.to(dev).eval()
    f16 = np.dtype(out_dtype) == np.float16
    cp = ChunkPredictor(model, n_range, patch_size, patch_overlap, batch_size, out_f16=f16)
    chunks = plan_chunks(start_ping, n_pings, preload_n_pings)
    if shard not in ("chunk", "patch"):
        raise ValueError(f"predict_survey: shard must be 'chunk' or 'patch', got {shard!r}")
    share_patches = shard == "patch"
    dist = torch.distributed
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if ordered_to_rank0 and share_patches:
        raise ValueError("predict_survey: ordered_to_rank0 belongs to shard='chunk' (with shard='patch' every rank already "
                         "yields every chunk)")
    ordered = multi and not share_patches and bool(ordered_to_rank0)
    all_chunks = chunks
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    if multi and not share_patches:
        chunks = chunks[rank::world]
    sender = ordered and rank != 0
    hand = _OrderedHandoff(rank, world, all_chunks, n_range, np.dtype(out_dtype)) if ordered else None
    if not chunks:
        return
    n_freq = len(segpipe.frequencies)
    widest = max(e - s for s, e in chunks)
    halo = patch_size[1]
    n_data = n_freq * (widest + 2 * halo) * n_range
    NS = 3                               # host staging slots: two chunks are being read while one is uploaded
    max_patches = 4 * (widest // (patch_size[0] - 2 * patch_overlap) + 2) * (n_range // (patch_size[1] - 2 * patch_overlap) + 2)
    n_misc = (widest + 2 * halo) + 4 * max_patches            # int32: seabed | global centres | slice-relative centres
    # staging (pinned host + device) is kept between surveys of the same geometry: page-locking 3 x 75 MB + the result
    # buffers costs 10-20 ms per call, a tenth of a 65536-ping survey
    key = (str(dev), n_data, widest, n_range, n_misc, f16)
    bufs = _STAGING.get(key)
    if bufs is None or bufs["busy"]:     # (busy: another generator over the same geometry is still running)
        fresh = {
            "stage_data": [torch.empty(n_data, dtype=torch.float32).pin_memory() for _ in range(NS)],
            "stage_lab": [torch.empty((widest, n_range), dtype=torch.int16).pin_memory() for _ in range(NS)],
            "stage_misc": [torch.empty(n_misc, dtype=torch.int32).pin_memory() for _ in range(NS)],
            "dev_misc": [torch.empty(n_misc, dtype=torch.int32, device=dev) for _ in range(2)],
            "dev_data": [torch.empty(n_data, dtype=torch.float32, device=dev) for _ in range(2)],
            "dev_lab": [torch.empty((widest, n_range), dtype=torch.int16, device=dev) for _ in range(2)],
            "pinned": [torch.empty(2 * n_range * widest, dtype=torch.float16 if f16 else torch.float32).pin_memory()
                       for _ in range(2)],
            "busy": False,
        }
        if bufs is None:
            _STAGING.clear()             # (one geometry at a time: the buffers are large)
            _STAGING[key] = fresh
        bufs = fresh
    bufs["busy"] = True
    stage_data, stage_lab, stage_misc = bufs["stage_data"], bufs["stage_lab"], bufs["stage_misc"]
    dev_misc = bufs["dev_misc"]
    uploaded = [torch.cuda.Event() for _ in range(NS)]        # host slot k may be overwritten once this has passed
    # device side: two resident chunk buffers; chunk i is uploaded on the copy stream while chunk i-1 computes
    dev_data, dev_lab = bufs["dev_data"], bufs["dev_lab"]
    computed = [torch.cuda.Event() for _ in range(2)]         # device slot may be overwritten once this has passed
    copy_stream = torch.cuda.Stream(device=dev)

    import time as _time
    tick = _time.perf_counter

    def note(key, t0):
        if stats is not None:
            stats.setdefault(key, []).append(tick() - t0)

    def fetch(i, s, e):
        t0 = tick()
        k = i % NS
        # ping extent of the data a patch of the chunk can touch (dataset.py:175-177): the patch columns depend on
        # (s, e) only, so the seabed of [lo, hi) -- which contains [s, e) -- is read ONCE (the zarr reader derives it from
        # the full 2-D mask every time, data_reader.py:864-865)
        xs = np.arange(s - (patch_overlap + 1), e - (patch_overlap + 1), patch_size[0] - 2 * patch_overlap) + patch_size[0] // 2
        lo = max(0, int(xs[0]) - patch_size[1] // 2)
        hi = min(n_pings, int(xs[-1]) + patch_size[1] // 2)
        sb = np.asarray(reader.get_seabed(lo, hi - lo, return_numpy=True)).astype(np.int32)
        seabed = sb[max(s - lo, 0):e - lo] if lo <= s else np.asarray(reader.get_seabed(s, e - s, return_numpy=True)).astype(np.int32)
        grid = plan_grid(n_range, int(seabed.max()), s, e, patch_size, patch_overlap)
        assert lo == max(0, int(grid[0, 1]) - patch_size[1] // 2) and hi == min(n_pings, int(grid[-1, 1]) + patch_size[1] // 2)
        uploaded[k].synchronize()                                        # (no-op until the slot has been used)
        data = reader.get_data_slice(idx_ping=lo, n_pings=hi - lo, frequencies=segpipe.frequencies,
                                     return_numpy=True)
        d_t = stage_data[k][:n_freq * (hi - lo) * n_range].view(n_freq, hi - lo, n_range)     # contiguous
        np.copyto(d_t.numpy(), data, casting="same_kind")
        l_t = None
        if labels_available:
            lab = reader.get_label_slice(idx_ping=s, n_pings=e - s, return_numpy=True)
            l_t = stage_lab[k][:e - s]
            np.copyto(l_t.numpy(), lab, casting="unsafe")
        # the seabed of every ping a patch of the chunk can touch (the scatter kernel evaluates the mask from it)
        sb, mask = seabed_vector_or_mask(reader, s, e, n_range, sb, lo)
        P = len(grid)
        assert (hi - lo) + 4 * P <= n_misc, "misc staging too small"
        m = stage_misc[k].numpy()
        m[:hi - lo] = sb
        cen = np.asarray(grid, dtype=np.int32)
        m[hi - lo:hi - lo + 2 * P] = cen.reshape(-1)
        loc = cen.copy()
        loc[:, 1] -= lo
        m[hi - lo + 2 * P:hi - lo + 4 * P] = loc.reshape(-1)
        note("fetch_s", t0)
        return grid, lo, hi, d_t, l_t, stage_misc[k][:hi - lo + 4 * P], mask

    pinned = bufs["pinned"]                      # flat: every chunk's [2, range, e - s] view of it is contiguous
    events = [torch.cuda.Event() for _ in range(2)]
    pending = None                      # (s, e, slot) of the chunk whose D2H copy is in flight
    main = torch.cuda.current_stream()
    def _loop():
        nonlocal pending
        with ThreadPoolExecutor(max_workers=2) as pool:
            futs = {j: pool.submit(fetch, j, *chunks[j]) for j in range(min(2, len(chunks)))}
            for i, (s, e) in enumerate(chunks):
                t0 = tick()
                grid, lo, hi, d_t, l_t, sb, mask = futs.pop(i).result()
                note("wait_fetch_s", t0)
                t0 = tick()
                if i + 2 < len(chunks):
                    futs[i + 2] = pool.submit(fetch, i + 2, *chunks[i + 2])
                slot = i & 1
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(computed[slot])                   # chunk i-2 is done with this device slot
                    d_d = dev_data[slot][:d_t.numel()].view(d_t.shape)
                    d_d.copy_(d_t, non_blocking=True)
                    l_d = None
                    if l_t is not None:
                        l_d = dev_lab[slot][:e - s]
                        l_d.copy_(l_t, non_blocking=True)
                    m_d = dev_misc[slot][:sb.numel()]
                    m_d.copy_(sb, non_blocking=True)                        # (sb: seabed | centres, pinned)
                    uploaded[i % NS].record()
                note("enq_upload_s", t0)
                t1 = tick()
                main.wait_stream(copy_stream)
                P = len(grid)
                if stats is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                if mask is None:
                    cp.load_chunk(d_d, lo, l_d, None, s, e, seabed=m_d[:hi - lo], seabed_ping0=lo)
                else:                                 # (a mask the vector rule cannot express: uploaded as it is)
                    cp.load_chunk(d_d, lo, l_d, mask, s, e)
                note("enq_load_s", t1)
                t1 = tick()
                out = cp.predict(grid, predict_fn=predict_fn, centres_dev=m_d[hi - lo:].view(2, P, 2),
                                 share_patches=share_patches)
                note("enq_predict_s", t1)
                if stats is not None:
                    ev1.record()
                    stats.setdefault("gpu_events", []).append((ev0, ev1))
                computed[slot].record()
                if sender:                            # ordered hand-off: the chunk goes to rank 0, nothing comes back here
                    hand.send(out)
                    note("enqueue_s", t0)
                    continue
                pinned[slot][:out.numel()].view(out.shape).copy_(out, non_blocking=True)
                events[slot].record()
                note("enqueue_s", t0)
                if pending is not None:
                    ps, pe, pslot = pending
                    t0 = tick()
                    events[pslot].synchronize()
                    note("wait_gpu_s", t0)
                    t0 = tick()
                    res = pinned[pslot][:2 * n_range * (pe - ps)].view(2, n_range, pe - ps).numpy().copy()
                    note("copy_out_s", t0)
                    yield ps, pe, res
                pending = (s, e, slot)
            if sender:
                hand.flush()
                return
            ps, pe, pslot = pending
            events[pslot].synchronize()
            yield ps, pe, pinned[pslot][:2 * n_range * (pe - ps)].view(2, n_range, pe - ps).numpy().copy()

    try:
        if ordered and rank == 0:
            yield from hand.merge(_loop())
        else:
            yield from _loop()
    finally:
        torch.cuda.current_stream().synchronize()     # (nothing of this survey still reads or writes the staging)
        bufs["busy"] = False


def predict_echogram_memm(echogram, segpipe, patch_size, patch_overlap, batch_size, predict_fn=None, meta_channels=None,
                          **kwargs):
    """``save_reader_predictions_memm`` (save_predict.py:222-265) for one memmap echogram: returns the
    ``[2, n_range, n_pings]`` float64 array the reference ``np.save``s (probabilities rounded to float16 first, :252).

    ``echogram``: the reference's ``Echogram`` API -- ``shape = (n_range, n_pings)``, ``data_memmaps(freqs)`` ->
    list of [n_range, n_pings] arrays, ``label_memmap()``, ``get_seabed(idx_ping, n_pings)``.  The whole echogram is
    one grid (ping_start 0); the arrays are transposed to the ping-major layout of the gather kernel on the GPU.
    """
    n_range, n_pings = echogram.shape
    model = segpipe.model.to(segpipe.device).eval()
    dev = segpipe.device
    seabed = np.asarray(echogram.get_seabed(0, n_pings)).astype(np.int32)
    grid = plan_grid(n_range, int(seabed.max()), 0, n_pings, patch_size, patch_overlap)
    if n_range <= patch_size[1]:
        grid = grid.copy()
        grid[:, 0] = n_range // 2            # get_crop_memmap (dataset.py:256-258): window covers the water column
    data = torch.stack([torch.as_tensor(np.ascontiguousarray(m, dtype=np.float32))
                        for m in echogram.data_memmaps(segpipe.frequencies)]).to(dev)
    data = data.permute(0, 2, 1).contiguous()                                     # [C, pings, range]
    labels = torch.as_tensor(np.ascontiguousarray(echogram.label_memmap()).astype(np.int16)).to(dev).t().contiguous()
    cp = ChunkPredictor(model, n_range, patch_size, patch_overlap, batch_size, out_f16=True)
    if meta_channels:                    # late metadata injection: the planes are built on the GPU, per batch of crops
        cp.meta_source = MetaSource.from_echogram(echogram, meta_channels, dev)
    cp.load_chunk(data, 0, labels, None, 0, n_pings, seabed=seabed, flavour="memm")
    out = cp.predict(grid, predict_fn=predict_fn)
    return out.cpu().numpy().astype(np.float64)
