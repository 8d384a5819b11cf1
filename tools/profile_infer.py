#!/usr/bin/env python3
"""Inference-only loop for rocprofv3 (eval forward + softmax, batch 32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import synth
m = pkg.UNet_Baseline(3, 4, precision=sys.argv[1] if len(sys.argv) > 1 else "bf16")
m.load_state_dict(synth.synth_state_dict(seed=0))
m.cuda().eval()
x = torch.from_numpy(synth.synth_echogram_batch(32, 4, 256, 256, seed=1)).cuda()
with torch.no_grad():
    for _ in range(6):
        m.predict_softmax(x)
torch.cuda.synchronize()
