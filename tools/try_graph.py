#!/usr/bin/env python3
"""Experiment: capture the fused training step in a HIP graph (torch.cuda.CUDAGraph) and compare step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth

B = 32
m = pkg.UNet_Baseline(3, 4, precision="bf16")
m.load_state_dict(synth.synth_state_dict(seed=0))
m = m.cuda()
eng = m.engine
x = torch.from_numpy(synth.synth_echogram_batch(B, 4, 256, 256, seed=1)).cuda()
lab = torch.from_numpy(synth.synth_labels(B, 256, 256, seed=2)).cuda()
cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")


def step():
    return eng.train_step(x, lab, cw, 0.005, 0.95)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("eager  %.3f ms/step" % timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
print("captured; loss tensor", tuple(loss.shape))
print("graph  %.3f ms/step" % timeit(g.replay))
print("loss after replays", float(loss))
