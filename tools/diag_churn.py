#!/usr/bin/env python3
"""Does host address-space churn slow the GPU step?  The fused step on a RESIDENT batch (eager, 40 steps, one sync at
the end) while a host thread pulls a DataLoader and (optionally) copies / releases its batches; nothing is uploaded."""
import os, sys, time, threading, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = pkg.UNet_Baseline(3, 4, precision=prec); m.load_state_dict(synth.synth_state_dict(seed=0)); m = m.cuda()
x = torch.from_numpy(synth.synth_echogram_batch(32, 4, 256, 256, seed=1)).cuda()
lab = torch.from_numpy(synth.synth_labels(32, 256, 256, seed=2)).cuda()
cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
m.engine.loss_scale_check_every = 0
LAG = int(os.environ.get("LAG", "0"))          # > 0: every iteration waits for the event recorded LAG steps back
def steps(n=40):
    for _ in range(3):
        m.engine.train_step(x, lab, cw, 0.005, 0.95)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event() for _ in range(n)]
    waited = 0.0
    t0 = time.perf_counter()
    for j in range(n):
        m.engine.train_step(x, lab, cw, 0.005, 0.95)
        evs[j].record()
        if LAG and j >= LAG:
            w0 = time.perf_counter()
            evs[j - LAG].synchronize()
            waited += time.perf_counter() - w0
    torch.cuda.synchronize()
    steps.waited = 1e3 * waited / n
    return 1e3 * (time.perf_counter() - t0) / n
print(f"{prec}: no host thread: {steps():.2f} ms/step", flush=True)
B = 32
data = np.random.rand(64, 4, 256, 256).astype(np.float32)
class DS(torch.utils.data.Dataset):
    def __len__(self): return 4000 * B
    def __getitem__(self, i): return {"data": data[i % 64]}
libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
pin2 = torch.empty(32, 4, 256, 256).pin_memory()
sys.setswitchinterval(2e-4)
import queue
dev2 = torch.empty(32, 4, 256, 256, device="cuda")
cs = torch.cuda.Stream()
for mode in ("memcpy+madvise", "memcpy+upload+madvise", "memcpy|madvise", "memcpy+upload|madvise", "memcpy+upload|del"):
    stop = [False]; cnt = [0]
    dead = queue.Queue()
    def release(b, xx):
        if "madvise" in mode:
            libc.madvise(xx.data_ptr(), xx.numel() * 4, 9)
        del b, xx
    def rel_thread():
        while True:
            it = dead.get()
            if it is None:
                return
            release(*it)
            it = None
    def churn():
        dl = torch.utils.data.DataLoader(DS(), batch_size=B, num_workers=4)
        for b in dl:
            xx = b["data"]
            np.copyto(pin2.numpy(), xx.numpy())
            if "upload" in mode:
                with torch.cuda.stream(cs):
                    dev2.copy_(pin2, non_blocking=True)
                cs.synchronize()
            if "|" in mode:
                dead.put((b, xx))
            else:
                release(b, xx)
            del b, xx
            cnt[0] += 1
            if stop[0]:
                break
        dead.put(None)
        del dl
    th = threading.Thread(target=churn, daemon=True); th.start()
    th2 = threading.Thread(target=rel_thread, daemon=True); th2.start()
    while cnt[0] < 5:
        time.sleep(0.05)
    c0, t0 = cnt[0], time.perf_counter()
    r = steps()
    rate = 1e3 * (time.perf_counter() - t0) / max(cnt[0] - c0, 1)
    stop[0] = True; th.join(); th2.join()
    print(f"{prec}: host thread [{mode}] ({rate:.1f} ms/batch): {r:.2f} ms/step (event wait {steps.waited:.2f} ms/step, LAG {LAG})", flush=True)
