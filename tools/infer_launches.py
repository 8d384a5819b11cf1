#!/usr/bin/env python3
"""Per-launch table of one inference forward (eval forward + softmax, batch 32): name, GFLOP, median us, TFLOP/s (executed).
usage: infer_launches.py [precision] [passes]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import hip, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "h3p"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
m = pkg.UNet_Baseline(3, 4, precision=prec, infer_precision=prec); m.load_state_dict(synth.synth_state_dict(seed=0)); m = m.cuda().eval()
x = torch.from_numpy(synth.synth_echogram_batch(32, 4, 256, 256, seed=1)).cuda()
with torch.no_grad():
    for _ in range(5):
        m.predict_softmax(x)
    torch.cuda.synchronize()
    hip.PROFILE = []
    for _ in range(n):
        m.predict_softmax(x)
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
