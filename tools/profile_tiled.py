#!/usr/bin/env python3
"""Where a chunk of predict_survey spends its time (BASELINE configs[3]): host stage timings + GPU kernel time."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth, tiled_inference as ti

n_pings = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dt = np.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else np.float32
reader = synth.SyntheticSurveyReader(n_pings=n_pings, n_range=1024, seabed_index=900, block=4096)
model = pkg.UNet_Baseline(3, 4, precision="bf16")
model.load_state_dict(synth.synth_state_dict(seed=0))
pipe = types.SimpleNamespace(model=model, device=torch.device("cuda"), frequencies=[18, 38, 120, 200])
for _ in ti.predict_survey(reader, pipe, (256, 256), 20, 32, 4096, start_ping=n_pings - 8192, out_dtype=dt):
    pass
torch.cuda.synchronize()
stats = {}
t0 = time.perf_counter()
n = 0
for s, e, out in ti.predict_survey(reader, pipe, (256, 256), 20, 32, 4096, out_dtype=dt, stats=stats):
    n += 95
dtot = time.perf_counter() - t0
print(f"{n} patches in {dtot*1e3:.1f} ms -> {n/dtot:.0f} patches/s, {dtot/(n/95)*1e3:.2f} ms per chunk, out {dt.__name__}")
torch.cuda.synchronize()
gpu = [a_.elapsed_time(b_) for a_, b_ in stats.pop("gpu_events")]
print("  GPU time per chunk in the pipeline (ms):", " ".join(f"{g:.2f}" for g in gpu))
for k, v in stats.items():
    print("   ", k, " ".join(f"{x*1e3:.1f}" for x in v))
for k, v in stats.items():
    print(f"  {k:14s} mean {np.mean(v)*1e3:7.2f} ms  max {np.max(v)*1e3:7.2f} ms  n {len(v)}")
# GPU-only time of one chunk's work (no host pipeline)
cp = ti.ChunkPredictor(model.cuda().eval(), 1024, (256, 256), 20, 32, out_f16=(dt == np.float16))
grid = ti.plan_grid(1024, 900, 0, 4096)
data = torch.from_numpy(reader.get_data_slice(0, 4096 + 128)).cuda()
lab = torch.from_numpy(reader.get_label_slice(0, 4096)).cuda()
sb = torch.from_numpy(reader.get_seabed(0, 4096 + 128).astype(np.int32)).cuda()
for _ in range(3):
    cp.load_chunk(data, 0, lab, None, 0, 4096, seabed=sb, seabed_ping0=0)
    cp.predict(grid)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    cp.load_chunk(data, 0, lab, None, 0, 4096, seabed=sb, seabed_ping0=0)
    cp.predict(grid)
b.record()
torch.cuda.synchronize()
print(f"GPU-resident chunk (gather + U-Net + softmax + scatter, 95 patches): {a.elapsed_time(b)/5:.2f} ms")
x = torch.from_numpy(synth.synth_echogram_batch(96, 4, 256, 256, seed=1)).cuda()
for B in (32, 96):
    xb = x[:B].contiguous()
    for _ in range(3):
        model.predict_softmax(xb)
    a.record()
    for _ in range(10):
        model.predict_softmax(xb)
    b.record()
    torch.cuda.synchronize()
    print(f"bare forward B={B}: {a.elapsed_time(b)/10:.2f} ms -> {B/(a.elapsed_time(b)/10)*1e3:.0f} patches/s")
