"""Tiled whole-survey inference on the GPU: the ``save_predict.py`` path of the reference.

Mirrors ``save_survey_predictions_zarr`` (crimac_unet/pipeline_train_predict/save_predict.py:137-220)
for the preload (zarr reader) flavour: the survey is cut into chunks of at most ``preload_n_pings``
pings (utils/preload_data_split.py:22-30), every chunk is gridded into overlapping patches
(batch/samplers/gridded.py:22-54), each patch is cropped + dB-transformed, pushed through the
U-Net + softmax, and the valid interior of its SANDEEL / OTHER probabilities is scattered into a
``[2, range, pings]`` array (save_predict.py:41-65).

Here the chunk is uploaded ONCE in the reader's own orientation and stays in HBM; crop, transform,
forward, softmax and scatter all run on the GPU (``crimac_gather_patches`` -> U-Net engine ->
``crimac_scatter_patches``); only the finished ``[2, range, pings]`` chunk comes back.  The
reference does the crop / transform / ``argwhere`` scatter per patch in numpy DataLoader workers and
moves 786 kB of softmax per patch over PCIe.

With ``torch.distributed`` initialised, patches of a chunk are dealt round-robin to the ranks
(``parallel.shard_indices``) and the per-rank outputs are summed (valid interiors are disjoint).
Writing zarr is out of scope: ``predict_survey`` yields numpy chunks that a
caller appends with the reference's ``create_xarray_ds_predictions`` / ``append_to_zarr``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip, parallel
from .hip import call, ptr

SEABED_PAD = 10        # mask_label_seabed.py:50-52
SEABED_MARGIN = 50     # gridded.py:150-156


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


class ChunkPredictor:
    """GPU state of one preloaded chunk: data, labels, seabed mask; gathers, predicts, scatters."""

    def __init__(self, model, n_range, patch_size=(256, 256), patch_overlap=20, batch_size=32):
        self.model = model
        self.engine = model.engine
        self.n_range = n_range
        self.patch_size = tuple(int(v) for v in patch_size)
        self.patch_overlap = int(patch_overlap)
        self.batch_size = int(batch_size)

    def load_chunk(self, data, data_ping0, labels, seabed_mask, start_ping, end_ping):
        """data [C, pings, range] fp32 (global ping of column 0 = data_ping0); labels [end-start, range]
        (or None); seabed_mask [end-start, range] uint8/bool for pings [start_ping, end_ping) (or None)."""
        dev = self.engine.device or next(self.model.parameters()).device
        self.data = torch.as_tensor(np.ascontiguousarray(data, dtype=np.float32)).to(dev)
        self.data_ping0 = int(data_ping0)
        self.labels = None if labels is None else torch.as_tensor(
            np.ascontiguousarray(labels).astype(np.int16)).to(dev)
        self.mask = None if seabed_mask is None else torch.as_tensor(
            np.ascontiguousarray(seabed_mask).astype(np.uint8)).to(dev)
        self.start_ping, self.end_ping = int(start_ping), int(end_ping)
        self.out = torch.zeros((2, self.n_range, self.end_ping - self.start_ping), dtype=torch.float32,
                               device=dev)

    def predict(self, grid, predict_fn=None):
        """Run all patches of ``grid`` ([P,2] global centres) and scatter them into ``self.out``.

        ``predict_fn(x_nhwc, P, H, W) -> probs [P,3,H,W]`` overrides the network (tests)."""
        eng = self.engine
        eng.bind()
        ph, pw = self.patch_size[1], self.patch_size[0]
        C = self.data.shape[0]
        world, rank, _ = (1, 0, 0)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        mine = parallel.shard_indices(len(grid), rank, world)
        for b0 in range(0, len(mine), self.batch_size):
            idx = mine[b0:b0 + self.batch_size]
            P = len(idx)
            cen = np.asarray(grid)[idx].astype(np.int32)
            local = cen.copy()
            local[:, 1] -= self.data_ping0
            cen_d = torch.from_numpy(np.ascontiguousarray(cen)).to(self.data.device)
            loc_d = torch.from_numpy(np.ascontiguousarray(local)).to(self.data.device)
            x = eng._buf("tiled.x", (P * ph * pw, 16))
            call("crimac_gather_patches", eng.prec, ptr(self.data), C, self.data.shape[1], self.n_range,
                 ptr(loc_d), P, ph, pw, ptr(x), 16)
            if predict_fn is None:
                probs = eng.forward_nhwc(x, P, ph, pw, training=False, softmax=True)
            else:
                probs = predict_fn(x, P, ph, pw)
            call("crimac_scatter_patches", ptr(probs), probs.shape[1], ptr(cen_d), P, ph, pw,
                 self.patch_overlap, self.start_ping, self.end_ping - self.start_ping, self.n_range,
                 ptr(self.labels), ptr(self.mask), self.start_ping, self.end_ping - self.start_ping,
                 ptr(self.data[0]), self.data_ping0, self.data.shape[1], SEABED_PAD, ptr(self.out))
        if world > 1:
            torch.distributed.all_reduce(self.out)      # interiors are disjoint: sum == union
        return self.out


def predict_survey(reader, segpipe, patch_size, patch_overlap, batch_size, preload_n_pings,
                   start_ping=0, labels_available=True, **kwargs):
    """Generator over chunks: yields ``(start_ping, end_ping, out[2, n_range, end-start] float32 numpy)``.

    ``reader``: the reference's zarr reader API (shape, get_data_slice, get_label_slice, get_seabed,
    get_seabed_mask); ``segpipe``: a ``SegPipeUNet`` with loaded parameters.

    Three stages overlap: a host thread reads chunk i+1 from ``reader`` (numpy / zarr I/O, no GPU calls)
    while the GPU gathers, predicts and scatters chunk i, and the result of chunk i-1 comes back through a
    pinned buffer (asynchronous D2H) before it is handed to the caller.
    """
    from concurrent.futures import ThreadPoolExecutor
    n_pings, n_range = reader.shape
    model = segpipe.model.to(segpipe.device).eval()
    cp = ChunkPredictor(model, n_range, patch_size, patch_overlap, batch_size)
    chunks = plan_chunks(start_ping, n_pings, preload_n_pings)

    def fetch(s, e):
        max_seabed = reader.get_seabed(s, e - s, return_numpy=False).max().values
        grid = plan_grid(n_range, max_seabed, s, e, patch_size, patch_overlap)
        lo = max(0, int(grid[0, 1]) - patch_size[1] // 2)              # dataset.py:175-177
        hi = min(n_pings, int(grid[-1, 1]) + patch_size[1] // 2)
        data = reader.get_data_slice(idx_ping=lo, n_pings=hi - lo, frequencies=segpipe.frequencies,
                                     return_numpy=True)
        labels = reader.get_label_slice(idx_ping=s, n_pings=e - s, return_numpy=True) if labels_available else None
        mask = np.asarray(reader.get_seabed_mask(s, e - s, 0, n_range, seabed_pad=0))
        return grid, lo, np.ascontiguousarray(data, dtype=np.float32), labels, mask

    widest = max(e - s for s, e in chunks)
    pinned = [torch.empty((2, n_range, widest), dtype=torch.float32).pin_memory() for _ in range(2)]
    events = [torch.cuda.Event() for _ in range(2)]
    pending = None                      # (s, e, slot) of the chunk whose D2H copy is in flight
    with ThreadPoolExecutor(max_workers=1) as pool:
        fut = pool.submit(fetch, *chunks[0])
        for i, (s, e) in enumerate(chunks):
            grid, lo, data, labels, mask = fut.result()
            if i + 1 < len(chunks):
                fut = pool.submit(fetch, *chunks[i + 1])
            cp.load_chunk(data, lo, labels, mask, s, e)
            out = cp.predict(grid)
            slot = i & 1
            pinned[slot][:, :, :e - s].copy_(out, non_blocking=True)
            events[slot].record()
            if pending is not None:
                ps, pe, pslot = pending
                events[pslot].synchronize()
                yield ps, pe, pinned[pslot][:, :, :pe - ps].numpy().copy()
            pending = (s, e, slot)
        ps, pe, pslot = pending
        events[pslot].synchronize()
        yield ps, pe, pinned[pslot][:, :, :pe - ps].numpy().copy()
