#!/usr/bin/env python3
"""Host time of one fused training step (the Python + ctypes enqueue of its ~130 launches, no synchronisation), and the
same step captured in a HIP graph (torch.cuda.CUDAGraph) and replayed.  usage: try_graph.py [precision ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth

B = 32
x = torch.from_numpy(synth.synth_echogram_batch(B, 4, 256, 256, seed=1)).cuda()
lab = torch.from_numpy(synth.synth_labels(B, 256, 256, seed=2)).cuda()
cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
for prec in (sys.argv[1:] or ["bf16", "h3f"]):
    m = pkg.UNet_Baseline(3, 4, precision=prec)
    m.load_state_dict(synth.synth_state_dict(seed=0))
    m = m.cuda()
    eng = m.engine
    eng.loss_scale_check_every = 0

    def step():
        return eng.train_step(x, lab, cw, 0.005, 0.95)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        host = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            h0 = time.perf_counter()
            fn()
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3

    e = timeit(step)
    print(f"{prec}: eager {e[0]:.3f} ms/step, host enqueue {e[1]:.3f} ms/step", flush=True)
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = step()
        r = timeit(g.replay)
        print(f"{prec}: graph {r[0]:.3f} ms/step, host replay {r[1]:.3f} ms/step; loss after replays {float(loss):.5f}", flush=True)
    except Exception as ex:
        print(f"{prec}: capture failed: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
    del m, eng
    torch.cuda.empty_cache()
