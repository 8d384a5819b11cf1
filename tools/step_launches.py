#!/usr/bin/env python3
"""Per-launch table of one serialized training step: name, algorithmic GFLOP, median us over N steps, TFLOP/s.
usage: step_launches.py [precision] [steps]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import hip, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = 32
m = pkg.UNet_Baseline(3, 4, precision=prec); m.load_state_dict(synth.synth_state_dict(seed=0)); m = m.cuda()
eng = m.engine
x = torch.from_numpy(synth.synth_echogram_batch(B, 4, 256, 256, seed=1)).cuda()
lab = torch.from_numpy(synth.synth_labels(B, 256, 256, seed=2)).cuda()
cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
eng.loss_scale_check_every = 0
eng.wgrad_side_streams, eng._side = 0, None
for _ in range(5):
    eng.train_step(x, lab, cw, 0.005, 0.95)
torch.cuda.synchronize()
hip.PROFILE = []
for _ in range(n):
    eng.train_step(x, lab, cw, 0.005, 0.95)
torch.cuda.synchronize()
prof, hip.PROFILE = hip.PROFILE, None
k = len(prof) // n
tot = 0.0
for p in range(k):
    ts = [prof[s * k + p][2].elapsed_time(prof[s * k + p][3]) for s in range(n)]
    med = statistics.median(ts)
    name, fl = prof[p][0], prof[p][1] or 0
    mf = prof[p][4] if len(prof[p]) > 4 else 1
    tot += med
    print(f"{p:3d} {name:34s} {fl / 1e9:9.1f} GFLOP x{mf}  {1e3 * med:8.1f} us  {fl * mf / (med * 1e-3) / 1e12 if med > 0 else 0:7.0f} TF(exec)")
print(f"sum of medians {tot:.3f} ms over {k} profiled launches")
