#!/usr/bin/env python3
"""Is the DataLoader-fed loop GIL-bound?  The staging thread (staging.BatchStager, 4 DataLoader workers) in front of
(a) a spin kernel, (b) the eager fused step on a resident batch, (c) the same step replayed from a HIP graph (the replay
call holds no GIL), (d) the eager step on the staged batch.  Prints ms per batch and the staging thread's phase times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import staging, synth

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, iters, nd = 32, int(os.environ.get("ITERS", "60")), 64
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), " affinity:", len(os.sched_getaffinity(0)), flush=True)
except Exception as e:
    print("cgroup:", e)
data = synth.synth_echogram_batch(nd, 4, 256, 256, seed=300)
labels = synth.synth_labels(nd, 256, 256, seed=301)


class DS(torch.utils.data.Dataset):
    def __len__(self):
        return iters * B

    def __getitem__(self, i):
        return {"data": data[i % nd], "labels": labels[i % nd], "center_coordinates": np.array([128, 128 + i], dtype=np.int64)}


dl = torch.utils.data.DataLoader(DS(), batch_size=B, num_workers=nw, drop_last=True, persistent_workers=nw > 0)
m = pkg.UNet_Baseline(3, 4, precision=prec)
m.load_state_dict(synth.synth_state_dict(seed=0))
m = m.cuda()
eng = m.engine
eng.loss_scale_check_every = 0
xr = torch.from_numpy(data[:B]).cuda()
lr_ = torch.from_numpy(labels[:B]).cuda()
cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
for _ in range(3):
    eng.train_step(xr, lr_, cw, 0.005, 0.95)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.train_step(xr, lr_, cw, 0.005, 0.95)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.train_step(xr, lr_, cw, 0.005, 0.95)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.train_step(xr, lr_, cw, 0.005, 0.95)
torch.cuda.synchronize()
print(f"{prec}: resident eager step {1e3 * (time.perf_counter() - t0) / 20:.2f} ms", flush=True)

modes = {
    "spin kernel" if not os.environ.get("TRACE") else "skip": lambda x, lab: torch.cuda._sleep(int(11.5e-3 * 2.1e9)),
    "eager step, resident batch": lambda x, lab: eng.train_step(xr, lr_, cw, 0.005, 0.95),
    "graph replay, resident batch": lambda x, lab: g.replay(),
    "eager step, staged batch": lambda x, lab: eng.train_step(x, lab, cw, 0.005, 0.95),
}
only = os.environ.get("ONLY")
for name, fn in modes.items():
    if only and only not in name:
        continue
    st = {}
    for timed in (False, True):
        st.clear()
        if os.environ.get("TRACE"):
            st["trace"] = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = 0.0
        for i, x, lab, b in staging.BatchStager(dl, "cuda:0", stats=st, yield_batch=False):
            h0 = time.perf_counter()
            fn(x, lab)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    tr = st.pop("trace", None)
    if tr:
        tr.sort(key=lambda e: e[2])
        base = tr[len(tr) // 2][2]
        for who, key, a, b in tr[len(tr) // 2: len(tr) // 2 + 60]:
            print(f"      {who:6s} {key:18s} {1e3 * (a - base):8.2f} -> {1e3 * (b - base):8.2f}  ({1e3 * (b - a):.2f})")
    print(f"{name}: {1e3 * dt / iters:.2f} ms/batch, host in step call {1e3 * host / iters:.2f}; per batch [ms]: "
          + ", ".join(f"{k[:-2]} {1e3 * v / iters:.2f}" for k, v in sorted(st.items())), flush=True)
