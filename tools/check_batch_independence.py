#!/usr/bin/env python3
"""Eval-mode patch independence at full size: logits of patches 9..10 inside a batch of 32 == the same two patches
alone, bit for bit (usage: check_batch_independence.py [precision] [start_filts]).  A context-dependent fp contraction
in the 1x1 head once broke this in the last bit."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_unet import make_model
from crimac_classifiers_unet_amd import synth
m = make_model(sys.argv[1] if len(sys.argv) > 1 else "fp16", start_filts=int(sys.argv[2]) if len(sys.argv) > 2 else 128).eval()
x = torch.from_numpy(synth.synth_echogram_batch(32, 4, 256, 256, seed=61)).cuda()
with torch.no_grad():
    for rep in range(3):
        full = m(x); part = m(x[9:11].contiguous()); full2 = m(x)
        d = (full[9:11] - part).abs()
        print("rep", rep, "full vs part max diff", float(d.max()), "n diff", int((d > 0).sum()), "full vs full2 equal", bool(torch.equal(full, full2)))
        if int((d > 0).sum()):
            idx = (d > 0).nonzero()
            print("  first diffs", idx[:5].tolist(), "rows", sorted(set(idx[:, 2].tolist()))[:10], "cols", sorted(set(idx[:, 3].tolist()))[:10])
