"""One-process-per-GPU data parallelism over RCCL / xGMI (the reference has no distributed code).

Training: every rank runs the same step on its own mini-batch shard; the only
exchange is the gradient all-reduce over the FLAT fp32 gradient buffer (31.04 M floats = 124 MB),
issued as a few large buckets (xGMI is point-to-point; large messages keep every link busy), started
during the backward pass as soon as a contiguous range of the buffer is final, and averaged by folding
1/world into the SGD kernel's ``grad_scale``.  BatchNorm statistics stay per-rank by default (torch DDP
default); ``sync_bn=True`` (engine / pipeline) all-reduces the per-channel sums on a process group of its own.

Inference: patches are independent; patch ``p`` of a chunk goes to rank ``p % world`` and the
per-rank softmax slabs are all-gathered.

Backend: ``nccl`` (= RCCL on ROCm) when the tensors are on a GPU, ``gloo`` on CPU (used only by the
CPU tests of this host logic).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    world, rank, local = env_world()
    if world == 1:
        return world, rank, local
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("CRIMAC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def bucket_bounds(n, bucket_elems):
    """[(start, end)] covering [0, n) in buckets of at most ``bucket_elems`` elements."""
    if bucket_elems <= 0:
        raise ValueError("bucket_elems must be positive")
    return [(s, min(s + bucket_elems, n)) for s in range(0, n, bucket_elems)]


class GradSync:
    """Bucketed all-reduce (sum) of the flat gradient buffer, overlapped with the backward pass.

    ``launch(flat_grad, lo, hi)`` may be called as soon as ``flat_grad[lo:hi]`` is final (the engine
    does so after the decoder and after the bottleneck encoder block, whose 26 M of the 31 M gradients
    are ready when 40-60 % of the backward pass is still to run); the collectives are issued
    asynchronously -- on the process group's own stream, ordered after the kernels already queued on
    the current stream -- and ``finish`` waits for them.  ``__call__`` reduces whatever has not been
    launched yet, waits for everything and returns the 1/world averaging scale (folded into SGD).
    """

    def __init__(self, bucket_mb=32.0, group=None, algo=None, force=False):
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.group = group
        # force: issue the collectives even in a process group of ONE rank (they are identities there).  A single GPU
        # can then drive every multi-rank code path of the step -- RCCL's stream-ordered waits, the chained
        # reduce-scatter + all-gather, the exchange launched from the side stream, the per-range optimiser step --
        # and must reproduce the plain single-rank step bit for bit (tests/test_gpu_unet.py)
        self.force = bool(force)
        # 'all_reduce' (default): one RCCL all-reduce per bucket.  'rs_ag': the same sum spelled out as
        # reduce-scatter + all-gather per bucket -- what RCCL's ring all-reduce does internally; the
        # explicit form lets the two halves be scheduled as separate collectives on the process group's stream.
        self.algo = algo or os.environ.get("CRIMAC_GRAD_EXCHANGE", "all_reduce")
        if self.algo not in ("all_reduce", "rs_ag"):
            raise ValueError(f"GradSync: unknown algo {self.algo!r}")
        self._works = []         # (range, work) in flight
        self._done = []          # [(lo, hi)] ranges already in flight this step
        self._shards = {}        # bucket -> reduce-scatter output buffer
        self._gathers = []       # (range, reduce-scatter work, bucket, shard) whose all-gather is still to be issued

    def world(self):
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def _stream_ordered(self):
        """RCCL ("nccl"): the collectives of one process group run on its own stream IN ISSUE ORDER, and ``wait()`` makes
        the current stream wait (no host block) -- an all-gather can be queued right behind its reduce-scatter.  gloo
        runs asynchronous collectives on worker threads without mutual order: the all-gather has to be issued after the
        reduce-scatter has completed (host wait)."""
        return dist.get_backend(self.group) == "nccl"

    def launch(self, flat_grad, lo, hi):
        if (self.world() == 1 and not self.force) or hi <= lo:
            return
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradSync(force=True) needs an initialised process group (world size 1 is enough)")
        for a, b in self._done:
            if lo < b and a < hi:
                raise RuntimeError(f"GradSync: range [{lo},{hi}) overlaps [{a},{b}) already in flight")
        self._done.append((lo, hi))
        world = self.world()
        rng = (lo, hi)
        for s, e in bucket_bounds(hi - lo, self.bucket_elems):
            buf = flat_grad[lo + s:lo + e]
            n = e - s
            if self.algo == "rs_ag" and n % world == 0 and n >= world:
                key = (lo + s, n)        # one shard per bucket: several buckets are in flight at once
                shard = self._shards.get(key)
                if shard is None or shard.device != buf.device:
                    shard = self._shards[key] = torch.empty(n // world, dtype=buf.dtype, device=buf.device)
                # the reduce-scatter starts now (overlapped with the rest of the backward pass)
                w = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self._stream_ordered():
                    # ... and its all-gather is chained behind it on the process group's stream: no host wait anywhere
                    self._works.append((rng, w))
                    self._works.append((rng, dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=True)))
                else:
                    self._gathers.append((rng, w, buf, shard))         # issued by finish_range() / finish()
            else:
                self._works.append((rng, dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def pending_ranges(self, n):
        """Complement of the launched ranges in [0, n)."""
        out, pos = [], 0
        for a, b in sorted(self._done):
            if a > pos:
                out.append((pos, a))
            pos = max(pos, b)
        if pos < n:
            out.append((pos, n))
        return out

    def finish_range(self, lo, hi):
        """Wait (RCCL: the current STREAM waits, the host does not) for the collectives of the launched range
        [lo, hi) only, so that the optimiser step of that range can run while later ranges are still being exchanged.
        Returns the 1/world averaging scale."""
        rng = (lo, hi)
        if rng not in self._done and (self.world() > 1 or self.force):
            raise RuntimeError(f"GradSync.finish_range: [{lo},{hi}) was not launched")
        keep = []
        for g in self._gathers:
            if g[0] != rng:
                keep.append(g)
                continue
            _, w, buf, shard = g
            w.wait()
            self._works.append((rng, dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=True)))
        self._gathers = keep
        rest = []
        for r, w in self._works:
            if r == rng:
                w.wait()
            else:
                rest.append((r, w))
        self._works = rest
        return 1.0 / self.world()

    def finish(self):
        for rng in list(self._done):
            self.finish_range(*rng)
        self._works, self._done, self._gathers = [], [], []
        return 1.0 / self.world()

    def __call__(self, flat_grad):
        if self.world() == 1 and not self.force:
            return 1.0
        for lo, hi in self.pending_ranges(flat_grad.numel()):
            self.launch(flat_grad, lo, hi)
        return self.finish()


def shard_indices(n_items, rank, world):
    """Indices of the items (patches) rank ``rank`` owns: round-robin ``p % world == rank``."""
    return list(range(rank, n_items, world))


def gather_shards(local, n_items, group=None):
    """All-gather per-rank result slabs back into item order.

    ``local``: [n_local, ...] results for ``shard_indices(n_items, rank, world)``; every rank gets
    the full [n_items, ...] tensor.  Ranks with one item fewer are padded for the collective.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_max = (n_items + world - 1) // world
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        out[idx] = parts[r][: len(idx)]
    return out


def all_reduce_scalars(t, group=None):
    """Sum a small tensor (loss numerator/denominator) over ranks in place."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
