"""Asynchronous input staging of the training loop (reference pipeline.py:161-164).

The reference moves every batch to the device inside the step (``batch['data'].float().to(device)``): the DataLoader
hands PAGEABLE tensors, so that copy is synchronous -- 37.7 MB (32 x 4 x 256 x 256 fp32) in front of a 12 ms step.
Here a host thread pulls the DataLoader one batch ahead and copies ``data`` and ``labels`` -- in the dtype they arrive in,
one plain memcpy each (numpy, GIL released; a torch CPU copy would start an OpenMP team per calling thread, one thread per
VISIBLE core: 256 on a GPU box whose cgroup grants 16 -- measured 43 ms per batch) -- into a ring of PINNED buffers; the
upload runs on a copy stream into a ring of device buffers, the reference's ``.float()`` happens on the device, and the
training stream waits for the upload's event only -- the H2D of step i + 1 runs under step i.
The same structure feeds the tiled-inference path (tiled_inference.predict_survey: reader thread -> pinned staging ->
copy stream -> two resident chunk buffers).
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch

RING = 4          # slots: one being filled by the host thread, one or two staged / uploading, one being consumed by the step


class _Slot:
    __slots__ = ("data_pin", "lab_pin", "data_dev", "lab_dev", "uploaded", "consumed", "used")

    def __init__(self):
        self.data_pin = self.lab_pin = self.data_dev = self.lab_dev = None
        self.uploaded = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.used = False


class BatchStager:
    """Iterate ``dataloader`` yielding ``(index, data_dev [B,C,H,W] float32, labels_dev [B,H,W], batch)`` with the
    tensors already (asynchronously) on ``device``; the caller's CURRENT stream is made to wait for the upload.

    The yielded device tensors belong to the ring: they stay valid until RING - 1 further batches have been yielded
    (the ring records an event on the caller's stream when the next batch is asked for, i.e. after the step that used
    them was enqueued)."""

    def __init__(self, dataloader, device, keys=("data", "labels"), stats=None):
        self.stats = stats                   # dict: seconds spent per phase (diagnostics: tools/diag_train_loop.py)
        self.dataloader = dataloader
        self.device = torch.device(device)
        self.keys = keys
        self.slots = [_Slot() for _ in range(RING)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._q = queue.Queue(maxsize=RING - 2)      # filled pinned slots waiting for their upload
        self._free = queue.Queue()
        for k in range(RING):
            self._free.put(k)
        self._stop = False
        self._err = None

    # -- host thread: DataLoader -> pinned ------------------------------------------------------------------------
    def _ensure(self, slot, data, labels):
        # The device buffers are allocated ON THE COPY STREAM: the caching allocator hands a block that was freed on a
        # stream back to allocations on that stream at once (later work of the stream is ordered behind its last use).
        # Allocated on the default stream, a ring buffer could be the block of a temporary the training stream's kernels
        # were still reading (the `.float()` copy of a float64 batch) -- and the upload, which runs on the copy stream,
        # is NOT ordered behind them: the first steps trained on half-overwritten crops (caught by
        # tests/test_gpu_unet.py::test_train_model_pinned_input_ring_equals_the_inline_copy with DataLoader workers).
        with torch.cuda.stream(self.copy_stream):
            if (slot.data_pin is None or slot.data_pin.shape != data.shape or slot.data_pin.dtype != data.dtype):
                slot.data_pin = torch.empty(data.shape, dtype=data.dtype).pin_memory()
                slot.data_dev = torch.empty(data.shape, dtype=data.dtype, device=self.device)
            if labels is not None and (slot.lab_pin is None or slot.lab_pin.shape != labels.shape
                                       or slot.lab_pin.dtype != labels.dtype):
                slot.lab_pin = torch.empty(labels.shape, dtype=labels.dtype).pin_memory()
                slot.lab_dev = torch.empty(labels.shape, dtype=labels.dtype, device=self.device)

    def _note(self, key, t0):
        if self.stats is not None:
            import time
            self.stats[key] = self.stats.get(key, 0.0) + time.perf_counter() - t0

    def _producer(self):
        import time
        try:
            it = iter(self.dataloader)
            i = -1
            while True:
                t0 = time.perf_counter()
                try:
                    batch = next(it)
                except StopIteration:
                    break
                i += 1
                self._note("producer_next_s", t0)
                t0 = time.perf_counter()
                k = self._free.get()
                self._note("producer_wait_slot_s", t0)
                if self._stop:
                    return
                slot = self.slots[k]
                data = batch[self.keys[0]]
                labels = batch.get(self.keys[1]) if len(self.keys) > 1 else None
                if not torch.is_tensor(data):
                    data = torch.as_tensor(data)
                if labels is not None and not torch.is_tensor(labels):
                    labels = torch.as_tensor(labels)
                if slot.used:
                    slot.uploaded.synchronize()          # the previous upload out of this pinned slot has finished
                self._ensure(slot, data, labels)
                t0 = time.perf_counter()
                np.copyto(slot.data_pin.numpy(), data.contiguous().numpy())      # memcpy, GIL released
                if labels is not None:
                    np.copyto(slot.lab_pin.numpy(), labels.contiguous().numpy())
                self._note("producer_copy_s", t0)
                t0 = time.perf_counter()
                self._q.put((i, k, batch, labels is not None))
                self._note("producer_wait_queue_s", t0)
                if self._stop:
                    return
        except BaseException as e:                        # surfaced in the consumer
            self._err = e
        finally:
            self._q.put(None)

    # -- consumer -------------------------------------------------------------------------------------------------
    # CPython hands the GIL from a running thread to a waiting one only every `sys.getswitchinterval()` = 5 ms unless the
    # running thread blocks.  The training thread spends a step's worth of host time in short Python stretches between
    # GIL-releasing launches; the staging thread, woken by the DataLoader's queue, needs the GIL a dozen times per batch
    # (unpickling the batch, wrapping tensors) and waited up to 5 ms each time: 14.2 ms per step for an 11.8 ms GPU step
    # (tools/diag_train_loop.py: staging thread 5.6 ms in next() + 3.7 ms memcpy + ~5 ms unaccounted per batch).  While
    # a stager runs the interval is 0.2 ms.
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
        th = threading.Thread(target=self._producer, daemon=True, name="crimac-batch-stager")
        th.start()
        main = torch.cuda.current_stream(self.device)
        prev = None
        try:
            import time
            while True:
                t0 = time.perf_counter()
                item = self._q.get()
                self._note("consumer_wait_batch_s", t0)
                if prev is not None:
                    # the step that used the previous slot has been enqueued on the caller's stream by now
                    self.slots[prev].consumed.record(main)
                    self._free.put(prev)
                    prev = None
                if item is None:
                    if self._err is not None:
                        raise self._err
                    return
                i, k, batch, has_lab = item
                slot = self.slots[k]
                with torch.cuda.stream(self.copy_stream):
                    if slot.used:
                        self.copy_stream.wait_event(slot.consumed)     # device slot free again
                    slot.data_dev.copy_(slot.data_pin, non_blocking=True)
                    if has_lab:
                        slot.lab_dev.copy_(slot.lab_pin, non_blocking=True)
                    slot.uploaded.record(self.copy_stream)
                slot.used = True
                main.wait_event(slot.uploaded)
                prev = k
                x = slot.data_dev
                if x.dtype != torch.float32:              # the reference's `.float()` (pipeline.py:163), on the device
                    x = x.float()
                yield i, x, (slot.lab_dev if has_lab else None), batch
        finally:
            self._stop = True
            try:                                          # unblock a producer waiting for a free slot
                self._free.put_nowait(0)
            except queue.Full:                            # pragma: no cover
                pass
            while th.is_alive():
                try:
                    self._q.get(timeout=0.05)
                except queue.Empty:
                    pass
            torch.cuda.current_stream(self.device).synchronize()     # nothing still reads the ring
