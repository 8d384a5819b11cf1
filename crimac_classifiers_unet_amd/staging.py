"""Asynchronous input staging of the training loop (reference pipeline.py:161-164).

The reference moves every batch to the device inside the step (``batch['data'].float().to(device)``): the DataLoader
hands PAGEABLE tensors, so that copy is synchronous -- 37.7 MB (32 x 4 x 256 x 256 fp32) in front of a 12 ms step.
Here two host threads work ahead of the training thread:
  stage    pulls the DataLoader (parent-side unpickling and mapping of the worker's shared-memory batch: 1-3 ms), copies
           ``data`` and ``labels`` -- in the dtype they arrive in, one plain memcpy each (numpy, GIL released; a torch
           CPU copy would start an OpenMP team per calling thread, one thread per VISIBLE core: 256 on a GPU box whose
           cgroup grants 16 -- measured 43 ms per batch) -- into a PINNED buffer (2.5-5 ms: first touch of the mapping),
           uploads it on a copy stream into a ring of device buffers and waits for the upload (0.75 ms);
  release  gives the batch's pages back (5-9 ms of page-table teardown, see ``release_pages``).
The reference's ``.float()`` happens on the device; the H2D of step i + 1 runs under step i.  No GPU event is waited for
or polled ACROSS threads (an event recorded by the training thread and queried from the staging thread read "not ready"
35-45 ms after a 0.6 ms upload had been issued, tools/diag_loop_graph.py round 4): the staging thread synchronises the
copy stream it issued on, and the training thread hands a device slot back after synchronising, itself, on the event it
recorded behind the step two iterations back -- which also keeps the host at most two steps ahead of the GPU.
Measured (round 4, 1 MI355X, 4 workers, 300 iterations): SegPipe.train_model 11.4 ms per bf16 step (resident batch 11.4)
and 17.8 ms per h3f step (17.9) -- the loop runs at the resident-batch rate.
The same structure feeds the tiled-inference path (tiled_inference.predict_survey: reader thread -> pinned staging ->
copy stream -> two resident chunk buffers).
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch

MADV_REMOVE = 9   # <linux/mman.h>: free the backing pages of a shared (tmpfs) mapping
_PAGE = 4096


def _libc_madvise():
    import ctypes
    libc = ctypes.CDLL(None, use_errno=True)
    libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    libc.madvise.restype = ctypes.c_int
    return libc.madvise


def collate_float32(samples):
    """``default_collate`` with the float64 ``data`` arrays of the samples cast to float32 BEFORE they are stacked.  The
    reference casts the collated batch (``batch['data'].float()``, pipeline.py:163,208): the same IEEE rounding of the same
    values, taken while the 2 MB crop is still in the worker's cache -- its zarr Dataset returns float64 crops
    (batch/dataset.py:361), i.e. 67 MB per batch of 32 through the stack, the shared-memory hand-over and the pinned copy
    instead of 33.  Samples that are not dicts, and every other key, go through ``default_collate`` untouched."""
    from torch.utils.data import default_collate
    if samples and isinstance(samples[0], dict) and "data" in samples[0]:
        def cast(v):
            if isinstance(v, np.ndarray) and v.dtype == np.float64:
                return v.astype(np.float32)
            if torch.is_tensor(v) and v.dtype == torch.float64:
                return v.float()
            return v
        samples = [{**s, "data": cast(s["data"])} for s in samples]
    return default_collate(samples)


def use_collate_float32(dataloader):
    """Swap a DataLoader's ``default_collate`` for ``collate_float32`` (before its first iterator exists: worker processes
    take the function when they start).  Returns True if swapped; any other collate function is left alone."""
    from torch.utils.data import default_collate
    if getattr(dataloader, "collate_fn", None) is not default_collate or getattr(dataloader, "batch_size", None) is None:
        return False
    if getattr(dataloader, "_iterator", None) is not None:          # (persistent workers already running the old function)
        return False
    dataloader.collate_fn = collate_float32
    return True


def collated_in_worker(dataloader):
    """True when every tensor the DataLoader yields is a FRESH shared-memory segment nobody else maps: worker processes
    + automatic batching + ``default_collate`` (or ``collate_float32`` above, which ends in it; it stacks the samples into storage it allocates in shared memory,
    torch/utils/data/_utils/collate.py) -- what the reference's four DataLoaders are (train.py:73,101, evaluate.py:69,107)."""
    from torch.utils.data import default_collate
    return (getattr(dataloader, "num_workers", 0) > 0 and getattr(dataloader, "batch_size", None) is not None
            and getattr(dataloader, "collate_fn", None) in (default_collate, collate_float32))


def release_shared_pages(batch, madvise=None, min_bytes=1 << 20):
    """Give the pages behind the shared-memory tensors of ``batch`` (a dict as default_collate builds it in a worker) back
    to the kernel with madvise(MADV_REMOVE) -- through ctypes, i.e. WITHOUT the GIL -- so that dropping the batch
    afterwards is cheap.  The tensors read as zeros from then on: only for batches nobody else refers to
    (``collated_in_worker``).  Returns the number of bytes released; a failing madvise only means the slow path."""
    madvise = madvise or _libc_madvise()
    done = 0
    for v in (batch.values() if isinstance(batch, dict) else ()):
        if torch.is_tensor(v) and v.device.type == "cpu" and v.is_shared() and v.is_contiguous():
            p, nb = v.data_ptr(), v.numel() * v.element_size()
            lo, hi = (p + _PAGE - 1) & ~(_PAGE - 1), (p + nb) & ~(_PAGE - 1)
            if hi - lo >= min_bytes and madvise(lo, hi - lo, MADV_REMOVE) == 0:
                done += hi - lo
    return done


COPY_THREADS = 4              # host threads of one pageable -> pinned batch copy
COPY_SPLIT_BYTES = 8 << 20    # arrays smaller than this are copied by the calling thread alone
_COPY_POOL = None


def parallel_copyto(dst, src):
    """``np.copyto(dst, src)`` for two C-contiguous arrays of one dtype, split along the first axis over COPY_THREADS
    threads (numpy releases the GIL inside the copy).  One thread moves a worker-collated batch at 9-10 GB/s (first touch of
    the shared-memory mapping): 3.6 ms for a float32 batch of 32 crops, 7.2 ms for the float64 crops of the reference's
    zarr flavour (batch/dataset.py:361) -- with the fetch and the upload in front and behind it the staging thread then
    needs longer than a 11.4 ms step (bench.py train_loop_raw, float64 leg: 0.89 of the resident-batch rate)."""
    global _COPY_POOL
    n = dst.shape[0] if dst.ndim else 0
    if dst.nbytes < COPY_SPLIT_BYTES or n < 2 or COPY_THREADS < 2:
        np.copyto(dst, src)
        return
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(max_workers=COPY_THREADS, thread_name_prefix="crimac-batch-copy")
    parts = min(COPY_THREADS, n)
    cuts = [n * k // parts for k in range(parts + 1)]
    futs = [_COPY_POOL.submit(np.copyto, dst[a:b], src[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    for f in futs:
        f.result()


RING = 6          # device / pinned slots: one being filled, up to three staged, two in steps the GPU may still be running
AHEAD = 2         # steps the training thread may be ahead of the GPU before it waits (frees the slot of step i - AHEAD)


class _Slot:
    __slots__ = ("data_pin", "lab_pin", "data_dev", "lab_dev", "done")

    def __init__(self):
        self.data_pin = self.lab_pin = self.data_dev = self.lab_dev = None
        self.done = torch.cuda.Event()       # recorded on the training stream behind the step that read the slot


# The ring survives its stager: a pass over the DataLoader (one call of train_model) used to start with six pinned
# allocations of 37 MB (hipHostMalloc: 20-70 ms each on a loaded host -- 0.4-1.3 ms per step averaged over 300 iterations,
# a third of the loop's distance to the resident-batch step).  One idle ring per device is kept here; it is reachable, so a
# forked DataLoader worker never finalises it (see _iterate).
_RING_LOCK = threading.Lock()
_IDLE_RINGS = {}


def _take_ring(device):
    with _RING_LOCK:
        ring = _IDLE_RINGS.pop(str(device), None)
    return ring if ring is not None else [_Slot() for _ in range(RING)]


def _give_ring(device, slots):
    with _RING_LOCK:
        _IDLE_RINGS.setdefault(str(device), slots)


def release_ring_cache():
    """Free the cached pinned / device buffers (they are reused by the next pass otherwise)."""
    with _RING_LOCK:
        _IDLE_RINGS.clear()


class BatchStager:
    """Iterate ``dataloader`` yielding ``(index, data_dev [B,C,H,W] float32, labels_dev [B,H,W], batch)`` with the
    tensors already on ``device`` (uploaded, and the upload complete, before they are yielded).

    The yielded device tensors belong to the ring: they stay valid until AHEAD further batches have been yielded (an event
    is recorded on the caller's CURRENT stream when the next batch is asked for, i.e. after the step that used them was
    enqueued; the slot is refilled only after that event has completed)."""

    def __init__(self, dataloader, device, keys=("data", "labels"), stats=None, yield_batch=True, release_pages=True):
        self.yield_batch = yield_batch       # False: `batch` is yielded as None (the loop needs the two tensors only)
        # Dropping a 37 MB batch a worker collated costs ~5-6 ms of page-table teardown + page freeing, and tensor
        # deallocation runs UNDER THE GIL: the training thread stalled that long every step (tools/diag_loop_graph.py:
        # 6.2-6.7 ms "release" per batch; a pure-Python ticker thread saw one > 1 ms gap per batch).  The release thread
        # frees the pages with madvise(MADV_REMOVE) through ctypes (no GIL) before the last reference goes; the
        # deallocation that follows finds nothing to tear down (0.6 ms).  Only for batches nobody else can see.
        import os
        # (release_pages=False -- yaml `release_batch_pages` -- or CRIMAC_STAGER_RELEASE_PAGES=0 keep the batches intact)
        self.release_pages = (bool(release_pages) and (not yield_batch) and collated_in_worker(dataloader)
                              and os.environ.get("CRIMAC_STAGER_RELEASE_PAGES", "1") != "0")
        self._madvise = _libc_madvise() if self.release_pages else None
        self._dead = queue.Queue()
        self.stats = stats                   # dict: seconds spent per phase (diagnostics: tools/diag_train_loop.py)
        self.dataloader = dataloader
        self.device = torch.device(device)
        self.keys = keys
        self.slots = _take_ring(self.device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._q = queue.Queue(maxsize=RING - AHEAD - 1)      # uploaded slots waiting for their step
        self._free = queue.Queue()                   # slots no GPU work refers to
        for k in range(RING):
            self._free.put(k)
        self._stop = False
        self._err = None
        self._it = None

    def _ensure(self, slot, data, labels):
        # The device buffers are allocated ON THE COPY STREAM: the caching allocator hands a block that was freed on a
        # stream back to allocations on that stream at once (later work of the stream is ordered behind its last use).
        # Allocated on the default stream, a ring buffer could be the block of a temporary the training stream's kernels
        # were still reading (the `.float()` copy of a float64 batch) -- and the upload, which runs on the copy stream,
        # is NOT ordered behind them: the first steps trained on half-overwritten crops (caught by
        # tests/test_gpu_unet.py::test_train_model_pinned_input_ring_equals_the_inline_copy with DataLoader workers).
        with torch.cuda.stream(self.copy_stream):
            if (slot.data_pin is None or slot.data_pin.shape != data.shape or slot.data_pin.dtype != data.dtype):
                slot.data_pin = torch.empty(data.shape, dtype=data.dtype, pin_memory=True)
                slot.data_dev = torch.empty(data.shape, dtype=data.dtype, device=self.device)
            if labels is not None and (slot.lab_pin is None or slot.lab_pin.shape != labels.shape
                                       or slot.lab_pin.dtype != labels.dtype):
                slot.lab_pin = torch.empty(labels.shape, dtype=labels.dtype, pin_memory=True)
                slot.lab_dev = torch.empty(labels.shape, dtype=labels.dtype, device=self.device)

    def _note(self, key, t0):
        if self.stats is not None:
            import time
            now = time.perf_counter()
            self.stats[key] = self.stats.get(key, 0.0) + now - t0
            tr = self.stats.get("trace")
            if tr is not None:
                tr.append((threading.current_thread().name[-5:], key[:-2], t0, now))

    # -- thread 1: DataLoader -> pinned memcpy -> upload ---------------------------------------------------------------------------
    def _stager(self):
        import time
        try:
            i = -1
            it = self._it
            while True:
                t0 = time.perf_counter()
                batch = next(it, None)
                self._note("fetch_next_s", t0)
                if batch is None or self._stop:
                    break
                i += 1
                t0 = time.perf_counter()
                k = self._free.get()
                self._note("stage_wait_slot_s", t0)
                if self._stop:
                    return
                slot = self.slots[k]
                t0 = time.perf_counter()
                data = batch[self.keys[0]]
                labels = batch.get(self.keys[1]) if len(self.keys) > 1 else None
                if not torch.is_tensor(data):
                    data = torch.as_tensor(data)
                if labels is not None and not torch.is_tensor(labels):
                    labels = torch.as_tensor(labels)
                self._note("stage_prep_s", t0)
                t0 = time.perf_counter()
                self._ensure(slot, data, labels)
                self._note("stage_ensure_s", t0)
                t0 = time.perf_counter()
                parallel_copyto(slot.data_pin.numpy(), data.contiguous().numpy())      # memcpy, GIL released
                if labels is not None:
                    np.copyto(slot.lab_pin.numpy(), labels.contiguous().numpy())
                self._note("stage_memcpy_s", t0)
                t0 = time.perf_counter()
                with torch.cuda.stream(self.copy_stream):
                    slot.data_dev.copy_(slot.data_pin, non_blocking=True)
                    if labels is not None:
                        slot.lab_dev.copy_(slot.lab_pin, non_blocking=True)
                self.copy_stream.synchronize()            # (GIL released) the batch is on the device from here on
                self._note("stage_upload_s", t0)
                has_lab = labels is not None
                t0 = time.perf_counter()
                del data, labels
                if not self.yield_batch:                  # the 37 MB mapping is released here, not by the training thread
                    self._dead.put(batch)
                    batch = None
                self._note("stage_handoff_s", t0)
                t0 = time.perf_counter()
                self._q.put((i, k, batch, has_lab))
                batch = None
                self._note("stage_wait_step_s", t0)
        except BaseException as e:
            self._err = e
        finally:
            self._q.put(None)

    # -- thread 2: page release -------------------------------------------------------------------------------------
    def _release(self, batch):
        import time
        t0 = time.perf_counter()
        if self.release_pages:
            release_shared_pages(batch, self._madvise)
        batch = None
        self._note("release_s", t0)

    def _releaser(self):
        while True:
            batch = self._dead.get()
            if batch is None:
                return
            self._release(batch)
            batch = None

    # -- consumer: the training thread ------------------------------------------------------------------------------
    # CPython hands the GIL from a running thread to a waiting one only every `sys.getswitchinterval()` = 5 ms unless the
    # running thread blocks.  The training thread spends a step's worth of host time in short Python stretches between
    # GIL-releasing launches; the staging threads, woken by the DataLoader's queue, need the GIL a dozen times per batch
    # (unpickling the batch, wrapping tensors) and waited up to 5 ms each time.  While a stager runs the interval is 0.2 ms.
    SWITCH_INTERVAL_S = 2e-4

    def __iter__(self):
        import sys
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, self.SWITCH_INTERVAL_S))
        try:
            yield from self._iterate()
        finally:
            sys.setswitchinterval(old_interval)

    def _iterate(self):
        import collections
        import time
        # The DataLoader's iterator is created HERE, by the training thread, before any helper thread exists: creating it
        # forks the worker processes, and a fork out of a helper thread of a process that has live threads of its own
        # (the HIP runtime's, a previous stager's) left workers that died with a segmentation fault now and then
        # (tests/test_gpu_unet.py::test_batch_stager_..., one run in three).  The staging thread only calls next().
        # ... and with no cyclic garbage pending: a forked worker starts with a garbage collection (multiprocessing's
        # after-fork hooks), and collecting an unreachable stager / engine of the PARENT there runs the destructors of its
        # events and pinned tensors -- HIP calls in a forked child: "Fatal Python error: Segmentation fault ...
        # Garbage-collecting ... _run_after_forkers" (seen with the stager of a loop that had ended in an exception).
        if getattr(self.dataloader, "num_workers", 0) > 0 and getattr(self.dataloader, "_iterator", None) is None:
            import gc                                      # (persistent workers that are already running: no fork)
            gc.collect()
        self._it = iter(self.dataloader)
        threads = [threading.Thread(target=self._stager, daemon=True, name="crimac-batch-stage")]
        rel = threading.Thread(target=self._releaser, daemon=True, name="crimac-batch-release")
        for th in threads + [rel]:
            th.start()
        main = torch.cuda.current_stream(self.device)
        in_flight = collections.deque()                   # slots whose step has been enqueued, oldest first
        prev = None
        try:
            while True:
                if prev is not None:
                    # the step that used the previous slot has been enqueued on the caller's stream by now
                    self.slots[prev].done.record(main)
                    in_flight.append(prev)
                    prev = None
                # slots go back to the staging thread BEFORE this thread waits for its next batch (the other order had
                # the two threads waiting for each other: 5-6 ms "wait slot" beside 2-4 ms "wait batch" per step)
                t0 = time.perf_counter()
                while len(in_flight) > AHEAD:             # recorded by THIS thread: completes when that step has
                    k_old = in_flight.popleft()
                    self.slots[k_old].done.synchronize()
                    self._free.put(k_old)
                while in_flight and self.slots[in_flight[0]].done.query():
                    self._free.put(in_flight.popleft())
                self._note("step_wait_gpu_s", t0)
                t0 = time.perf_counter()
                item = self._q.get()
                self._note("step_wait_batch_s", t0)
                if item is None:
                    if self._err is not None:
                        err, self._err = self._err, None
                        raise err
                    return
                i, k, batch, has_lab = item
                slot = self.slots[k]
                prev = k
                x = slot.data_dev
                if x.dtype != torch.float32:              # the reference's `.float()` (pipeline.py:163), on the device
                    x = x.float()
                yield i, x, (slot.lab_dev if has_lab else None), batch
        finally:
            self._stop = True
            try:                                          # unblock a stager waiting for a free slot
                self._free.put_nowait(0)
            except queue.Full:                            # pragma: no cover
                pass
            while any(th.is_alive() for th in threads):   # drain the queue until the staging thread has left
                try:
                    self._q.get(timeout=0.02)
                except queue.Empty:
                    pass
            self._dead.put(None)
            rel.join()
            self._it = None                                # (the workers are shut down by this thread, too)
            torch.cuda.current_stream(self.device).synchronize()     # nothing still reads the ring
            # the ring's events, pinned and device buffers leave the stager NOW (back to the module's cache, where they stay
            # reachable) -- not whenever a garbage collection finds the stager: the exception kept in _err refers to the
            # staging thread's frame, which refers to the stager (a cycle), and the collection might be a forked worker's
            err, self._err = self._err, None
            slots, self.slots = self.slots, []
            _give_ring(self.device, slots)
            self.copy_stream = None
            del err, slots
