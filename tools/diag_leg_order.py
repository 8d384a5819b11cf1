#!/usr/bin/env python3
"""Does a training measurement depend on what the process ran before it?  Steps of one precision after another in ONE
process, in the given order, each on a fresh model (previous one deleted, allocator cache emptied -- or not).
usage: diag_leg_order.py [--keep-cache] prec1 prec2 ...   (a precision may be 'tiled:<prec>' for a tiled-inference pass)"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth, tiled_inference as ti

keep = "--keep-cache" in sys.argv
legs = [a for a in sys.argv[1:] if not a.startswith("--")]
dev = torch.device("cuda", 0)
x = torch.from_numpy(synth.synth_echogram_batch(32, 4, 256, 256, seed=100)).to(dev)
lab = torch.from_numpy(synth.synth_labels(32, 256, 256, seed=200)).to(dev)
cw = torch.tensor([10.0, 300.0, 250.0], device=dev)
for leg in legs:
    if leg.startswith("tiled:"):
        prec = leg.split(":")[1]
        m = pkg.UNet_Baseline(3, 4, precision=prec); m.load_state_dict(synth.synth_state_dict(seed=0)); m.to(dev).eval()
        reader = synth.SyntheticSurveyReader(n_pings=32768, n_range=1024, seabed_index=900, block=4096)
        pipe = types.SimpleNamespace(model=m, device=dev, frequencies=[18, 38, 120, 200])
        t0 = time.perf_counter()
        for _ in ti.predict_survey(reader, pipe, (256, 256), 20, 32, 4096, out_dtype=np.float16):
            pass
        torch.cuda.synchronize()
        print(f"{leg}: {time.perf_counter() - t0:.2f} s", flush=True)
        ti.release_staging()
        del m, reader, pipe
    else:
        m = pkg.UNet_Baseline(3, 4, precision=leg); m.load_state_dict(synth.synth_state_dict(seed=0)); m.to(dev).train()
        eng = m.engine
        for _ in range(5):
            eng.train_step(x, lab, cw, lr=0.005, momentum=0.95)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.train_step(x, lab, cw, lr=0.005, momentum=0.95)
        torch.cuda.synchronize()
        print(f"{leg}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step  (reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB)", flush=True)
        del m, eng
    if not keep:
        torch.cuda.empty_cache()
