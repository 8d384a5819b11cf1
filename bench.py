#!/usr/bin/env python3
"""Benchmark of the U-Net hot path on MI355X (BASELINE.json: echogram patches/sec, 4ch 256x256).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic echogram patches already resident
in HBM: the full training step of the reference loop (forward + weighted CE + backward + SGD,
pipeline.py:163-178), batch 32 per GPU (BASELINE.json configs[1]); for N>1 the mini-batches shard
over ranks and gradients are all-reduced over RCCL every step (configs[2], weak scaling).
The inference rate (eval forward + softmax, pipeline.py:205-219) is measured next to it and
reported in the same JSON line as ``infer_patches_per_s``.

``roofline`` (per-kernel HIP-event durations; taken in a serialized pass right after the timed region because
the timed region runs the weight gradients concurrently on a side stream): the dominant kernel is the halo-staged implicit-GEMM 3x3 convolution (crimac_conv3x3 =
conv3x3_wch_kernel / conv3x3_p64_kernel / conv3x3_glds_w4_kernel / conv3x3_c16_kernel by layer shape): algorithmic
FLOPs of its launches (2*taps*Cin*N*M each) / their HIP-event durations, measured
inside the timed region, against the dense bf16 MFMA peak (2.5 PFLOP/s).
``cpu_baseline``: the CPU oracle (oracle/unet_oracle.py, kind "port") timed on this box's host
cores on a bounded sample (batch 2, a few steps), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_PATCH = 96.43      # SURVEY.md §8(a) a10, hook-counted on the reference module
TRAIN_GFLOP_PER_PATCH = 288.98
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32x3": 2500.0 / 3.0, "f32x6": 2500.0 / 6.0}   # dense bf16 / MFMAs per product
DTYPE_LABEL = {"bf16": "bf16", "f32x3": "fp32 storage, 3x bf16 MFMA per product",
               "f32x6": "fp32 storage, 6x bf16 MFMA per product (fp32-equivalent)"}


def cpu_baseline(batch=2, iters=3):
    """Time the oracle's training step and eval forward on the host cores (bounded sample)."""
    from crimac_classifiers_unet_amd import synth
    from oracle import unet_oracle as orc
    # host cores actually usable: affinity mask, capped at the GPU box's CPU share (16 per GPU);
    # oversubscribing a cgroup-limited box with one thread per visible core stalls for minutes
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("CRIMAC_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline on {cores} threads (os.cpu_count()={os.cpu_count()})", file=sys.stderr, flush=True)
    sd = synth.synth_state_dict(seed=0)
    x = torch.from_numpy(synth.synth_echogram_batch(batch, 4, 256, 256, seed=1))
    lab = torch.from_numpy(synth.synth_labels(batch, 256, 256, seed=2))
    vel = {}
    orc.loss_and_grads(sd, x, lab)        # warm-up
    t0 = time.perf_counter()
    state = sd
    for _ in range(iters):
        loss, _, grads, stats = orc.loss_and_grads(state, x, lab)
        state, vel = orc.sgd_momentum_step(dict(state), grads, vel, 0.005, 0.95)
        state.update(stats)
    t_train = (time.perf_counter() - t0) / iters
    orc.predict(sd, x)
    t0 = time.perf_counter()
    for _ in range(iters):
        orc.predict(sd, x, return_softmax=True)
    t_inf = (time.perf_counter() - t0) / iters
    return {"value": batch / t_train, "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": f"oracle train step, batch {batch} x 4x256x256 fp32, {iters} iters after 1 warm-up",
            "infer_value": batch / t_inf}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32x3", "f32x6"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    args = ap.parse_args()

    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import parallel, synth, hip

    world, rank, local = parallel.init_distributed()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    local = local % max(torch.cuda.device_count(), 1)     # (rehearsals with several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    model = pkg.UNet_Baseline(3, 4, precision=args.precision)
    model.load_state_dict(synth.synth_state_dict(seed=0))      # random-init weights of the architecture
    model.to(dev).train()
    eng = model.engine
    B = args.batch
    x = torch.from_numpy(synth.synth_echogram_batch(B, 4, 256, 256, seed=100 + rank)).to(dev)
    lab = torch.from_numpy(synth.synth_labels(B, 256, 256, seed=200 + rank)).to(dev)
    cw = torch.tensor([10.0, 300.0, 250.0], device=dev)
    grad_sync = parallel.GradSync()

    def step():
        return eng.train_step(x, lab, cw, lr=0.005, momentum=0.95, grad_sync=grad_sync)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log(f"model ready on {dev}, precision {args.precision}, batch {B}/GPU, world {world}")
    for _ in range(args.warmup):
        loss = step()
    barrier()
    log("warm-up done")
    # HIP-event instrumentation of the dominant kernel (igemm conv launches) inside the timed region
    hip.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {1e3 * elapsed / args.steps:.2f} ms/step")
    prof_timed, hip.PROFILE = hip.PROFILE, None
    final_loss = float(loss)                    # (loss of the last timed step)
    # Per-kernel durations: in the timed region the weight gradients run on a side stream CONCURRENTLY with the
    # input-gradient convolutions (engine._wgrad), so the HIP-event intervals of the two overlap and each contains
    # the other's share of the GPU.  The roofline figures therefore come from a second, serialized pass (same
    # steps, side stream off) right after the timed region; the overlapped averages are reported next to them.
    prof = prof_timed
    serialized = False
    if eng.wgrad_side_streams > 0:
        saved_cfg = (eng.wgrad_side_streams, eng._side)
        eng.wgrad_side_streams, eng._side = 0, None
        for _ in range(2):
            step()
        barrier()
        hip.PROFILE = []
        for _ in range(max(3, min(args.steps, 10))):
            loss = step()
        barrier()
        prof, hip.PROFILE = hip.PROFILE, None
        eng.wgrad_side_streams, eng._side = saved_cfg
        serialized = True
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    assert final_loss == final_loss, "training diverged (NaN loss)"

    # dominant-kernel roofline
    def kernel_rate(kname, records=None):
        sel = [(f, s.elapsed_time(e)) for n, f, s, e in (prof if records is None else records) if n == kname]
        fl, ms_ = sum(f for f, _ in sel), sum(m for _, m in sel)
        return (fl / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0), ms_, len(sel)

    achieved, ms, n_launch = kernel_rate("crimac_conv3x3")
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    # (tools/pmc_traffic.py: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); bench.py
    # itself cannot collect counters
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
        sel = [v for k, v in pmc.items() if k.startswith("conv3x3")]
        nl = sum(v["launches"] for v in sel)
        traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / nl if nl else None
    except Exception:
        pass
    wg_achieved, wg_ms, wg_n = kernel_rate("crimac_wgrad")
    peak = MFMA_PEAK_TFLOPS[args.precision]

    infer = None
    if not args.no_infer:
        model.eval()
        with torch.no_grad():
            for _ in range(max(args.warmup // 2, 2)):
                model.predict_softmax(x)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                model.predict_softmax(x)
            barrier()
            ti = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([ti], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ti = float(t)
        infer = world * B * args.steps / ti

    if rank == 0:
        value = world * B * args.steps / elapsed
        out = {
            "metric": "echogram patches/sec (4ch 256x256), training step (fwd+weighted CE+bwd+SGD)",
            "value": value, "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.precision],
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: U-Net (depth 5, 64 filters) train step, "
                                   "batch 32 x 4x256x256 per GPU" + (
                                       f", data-parallel over {world} GPUs (RCCL all-reduce)" if world > 1 else ""),
                       "global_batch": world * B, "patch": [4, 256, 256], "precision": args.precision,
                       "parallelism": f"dp{world}"},
            "train_tflops": value * TRAIN_GFLOP_PER_PATCH / 1e3,
            "infer_patches_per_s": infer,
            "infer_tflops": infer * FWD_GFLOP_PER_PATCH / 1e3 if infer else None,
            "final_loss": final_loss,
            "roofline": {"bound": "mfma", "kernel": "crimac_conv3x3: conv3x3_wch_kernel + conv3x3_p64_kernel + conv3x3_glds_w4_kernel + conv3x3_c16_kernel "
                                                           "(halo-staged implicit-GEMM 3x3 conv, fwd + dgrad, all layers)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_note": "HBM (L2-miss) bytes per launch, rocprofv3 PMC, profiles/r01_pmc_traffic.json",
                         "algorithmic_flops_per_launch": (achieved * 1e12) * (ms * 1e-3) / max(n_launch, 1),
                         "launches": n_launch, "avg_launch_us": 1e3 * ms / max(n_launch, 1),
                         "measured": ("serialized pass after the timed region (weight-gradient side stream off); "
                                      "in the timed region these launches overlap the weight gradients"
                                      if serialized else "timed region"),
                         "avg_launch_us_timed_region_overlapped": (lambda r: 1e3 * r[1] / max(r[2], 1))(
                             kernel_rate("crimac_conv3x3", prof_timed))},
            "roofline_wgrad": {"bound": "mfma", "kernel": "wgrad_kernel (weight gradient, all shapes)",
                               "achieved": wg_achieved, "peak": peak, "unit": "TFLOP/s",
                               "frac": wg_achieved / peak, "traffic": None, "launches": wg_n,
                               "avg_launch_us": 1e3 * wg_ms / max(wg_n, 1),
                               "avg_launch_us_timed_region_overlapped": (lambda r: 1e3 * r[1] / max(r[2], 1))(
                                   kernel_rate("crimac_wgrad", prof_timed))},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
