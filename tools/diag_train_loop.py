#!/usr/bin/env python3
"""Where the time of SegPipe.train_model goes when a DataLoader feeds it (bench.py's train_loop leg): seconds the staging
thread waits for the DataLoader, copies into pinned memory, waits for a free slot / the consumer; seconds the training
thread waits for a staged batch; per precision and worker count.  usage: diag_train_loop.py [precision] [workers ...]"""
import contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crimac_classifiers_unet_amd as pkg
from crimac_classifiers_unet_amd import staging, synth

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
workers = [int(a) for a in sys.argv[2:]] or [4, 8]
B, iters, nd = 32, int(os.environ.get("ITERS", "60")), 64
data = synth.synth_echogram_batch(nd, 4, 256, 256, seed=300)
labels = synth.synth_labels(nd, 256, 256, seed=301)


class DS(torch.utils.data.Dataset):
    def __len__(self):
        return iters * B

    def __getitem__(self, i):
        return {"data": data[i % nd], "labels": labels[i % nd], "center_coordinates": np.array([128, 128 + i], dtype=np.int64)}


def loader(nw):
    return torch.utils.data.DataLoader(DS(), batch_size=B, num_workers=nw, drop_last=True, persistent_workers=nw > 0)


# (a) the DataLoader alone, (b) + memcpy into pinned memory in the same thread, (c) the stager in front of a dummy step
#     (one GPU spin kernel of about a bf16 step, no host work)
pin = torch.empty(B, 4, 256, 256).pin_memory()
for nw in (workers if os.environ.get('ALONE') else []):
    dl = loader(nw)
    for what in ("next", "next+memcpy"):
        for timed in (False, True):
            t0 = time.perf_counter()
            for b in dl:
                if what != "next":
                    np.copyto(pin.numpy(), b["data"].numpy())
            dt = time.perf_counter() - t0
        print(f"DataLoader alone, workers {nw}, {what}: {1e3 * dt / iters:.2f} ms/batch", flush=True)
    st = {}
    for timed in (False, True):
        st.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, x, lab, b in staging.BatchStager(dl, "cuda:0", stats=st):
            torch.cuda._sleep(int(11.5e-3 * 2.1e9))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"stager + 11.5 ms spin kernel, workers {nw}: {1e3 * dt / iters:.2f} ms/batch; per batch [ms]: "
          + ", ".join(f"{k[:-2]} {1e3 * v / iters:.2f}" for k, v in sorted(st.items())), flush=True)
    del dl

stats = {}
for nw in workers:
    dl = loader(nw)
    pipe = pkg.SegPipeUNet(checkpoint_dir=None, data_mode="zarr", frequencies=[18, 38, 120, 200], patch_size=[256, 256],
                           loss_type="CE", lr=0.005, lr_reduction=0.5, lr_step=1000, momentum=0.95, batch_size=B, num_workers=nw,
                           iterations=iters, test_iter=10, log_step=10 ** 9, save_model_params=False, meta_channels=[],
                           late_meta_inject=False, eval_mode="all", experiment_name="diag", precision=prec, infer_precision=prec,
                           loss_flush=10 ** 9)
    pipe.model.load_state_dict(synth.synth_state_dict(seed=0))
    pipe.stager_stats = stats
    with contextlib.redirect_stdout(sys.stderr):
        pipe.train_model(dl, None, None)
        torch.cuda.synchronize()
        stats.clear()
        t0 = time.perf_counter()
        pipe.train_model(dl, None, None)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{prec} workers {nw}: {1e3 * dt / iters:.2f} ms/step, {iters * B / dt:.0f} patches/s; per step [ms]: "
          + ", ".join(f"{k[:-2]} {1e3 * v / iters:.2f}" for k, v in sorted(stats.items())))
    del dl, pipe
