#!/usr/bin/env python3
"""Pinned host -> device upload of one batch (33.5 MB): alone, under a spin kernel on another stream, under the fused step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pin = torch.empty(32, 4, 256, 256).pin_memory()
dev = torch.empty(32, 4, 256, 256, device="cuda")
cs = torch.cuda.Stream()
def upload_ms(n=10, under=None):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        if under:
            under()
        t0 = time.perf_counter()
        with torch.cuda.stream(cs):
            dev.copy_(pin, non_blocking=True)
            e = torch.cuda.Event(); e.record(cs)
        t1 = time.perf_counter()
        e.synchronize()
        ts.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
    ts.sort(key=lambda t: t[1])
    return ts[len(ts) // 2]
print("alone: issue %.3f ms, done after %.3f ms" % upload_ms())
print("under a 11.5 ms spin kernel: issue %.3f ms, done after %.3f ms" % upload_ms(under=lambda: torch.cuda._sleep(int(11.5e-3 * 2.1e9))))
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth
m = pkg.UNet_Baseline(3, 4, precision="bf16"); m.load_state_dict(synth.synth_state_dict(seed=0)); m = m.cuda()
x = torch.from_numpy(synth.synth_echogram_batch(32, 4, 256, 256, seed=1)).cuda()
lab = torch.from_numpy(synth.synth_labels(32, 256, 256, seed=2)).cuda()
cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
for _ in range(3):
    m.engine.train_step(x, lab, cw, 0.005, 0.95)
print("under the fused bf16 step: issue %.3f ms, done after %.3f ms" % upload_ms(under=lambda: m.engine.train_step(x, lab, cw, 0.005, 0.95)))

# the same upload while a host thread churns the address space the way the staging threads do
import threading, ctypes
import numpy as np
B = 32
data = np.random.rand(64, 4, 256, 256).astype(np.float32)
class DS(torch.utils.data.Dataset):
    def __len__(self): return 4000 * B
    def __getitem__(self, i): return {"data": data[i % 64]}
libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
pin2 = torch.empty(32, 4, 256, 256).pin_memory()
for mode in ("memcpy+del", "memcpy+madvise", "pread+del", "pread+madvise", "next only (+del)"):
    stop = [False]
    cnt = [0]
    def churn():
        dl = torch.utils.data.DataLoader(DS(), batch_size=B, num_workers=4)
        for b in dl:
            x = b["data"]
            if mode.startswith("memcpy"):
                np.copyto(pin2.numpy(), x.numpy())
            elif mode.startswith("pread"):
                fd, size = x.untyped_storage()._share_fd_cpu_()
                os.preadv(fd, [memoryview(pin2.numpy()).cast("B")], 0)
            if mode.endswith("madvise"):
                libc.madvise(x.data_ptr(), x.numel() * 4, 9)
            del b, x
            cnt[0] += 1
            if stop[0]:
                break
        del dl
    th = threading.Thread(target=churn, daemon=True)
    th.start()
    while cnt[0] < 5:
        time.sleep(0.05)
    c0, t0 = cnt[0], time.perf_counter()
    r = upload_ms(n=40, under=lambda: m.engine.train_step(x, lab, cw, 0.005, 0.95))
    rate = (time.perf_counter() - t0) / max(cnt[0] - c0, 1)
    stop[0] = True
    th.join()
    print(f"under the fused step + host thread [{mode}] ({1e3 * rate:.1f} ms/batch): issue %.3f ms, done after %.3f ms" % r, flush=True)
