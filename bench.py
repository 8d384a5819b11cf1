#!/usr/bin/env python3
"""Benchmark of the U-Net hot path on MI355X (BASELINE.json: echogram patches/sec, 4ch 256x256).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: this process spawns N ranks itself (one per GPU, RCCL) BEFORE touching
the GPU and prints rank 0's JSON line; under ``python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`` it is one of the N ranks (RANK / LOCAL_RANK / WORLD_SIZE from the environment).

A "step" is one pass of the hot path over one batch of synthetic echogram patches already resident in HBM: the
full training step of the reference loop (forward + weighted CE + backward + SGD, pipeline.py:163-178), batch 32
per GPU (BASELINE.json configs[1]); for N > 1 the mini-batches shard over ranks and gradients are exchanged
over RCCL every step (configs[2], weak scaling).  The JSON line also carries
  * ``infer_patches_per_s``   eval forward + softmax (pipeline.py:205-219) on the same batch;
  * ``parity_mode``           the same two measurements in a precision that meets the north-star parity bar
                              (h3p: fp16 plane pairs, 3 MFMAs per product, ~2^-21: <= 1e-3 rel on logits and identical
                              argmax masks -- measured in this run on the reference's golden crop, ``golden_parity``;
                              --parity-precision f32x6 for the 6-MFMA mode), with its roofline;
  * ``tiled``                 BASELINE configs[3]: tiled whole-survey inference (save_predict.py path), synthetic
                              survey 4 x 65536 pings x 1024 range, preload_n_pings 4096, host reader + H2D + crop/dB +
                              U-Net + softmax + scatter + D2H all inside the timed region;
  * ``roofline``              dominant kernel (3x3 implicit-GEMM convolution, all launches): algorithmic FLOPs
                              (2*9*Cin*Cout*B*H*W per launch) / HIP-event durations taken in a
                              serialized pass right after the timed region (the timed region overlaps the weight
                              gradients on a side stream), against the dense bf16 MFMA peak of 2.5 PFLOP/s;
  * ``cpu_baseline``          the CPU oracle (oracle/unet_oracle.py, kind "port") on this box's host cores,
                              bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_PATCH = 96.43      # SURVEY.md §8(a) a10, hook-counted on the reference module (start_filts 64)
TRAIN_GFLOP_PER_PATCH = 288.98
# dense 16-bit MFMA peak / MFMAs per product of the mode
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "f32x3": 2500.0 / 3.0, "f32h3": 2500.0 / 3.0, "h3p": 2500.0 / 3.0,
                    "f32x6": 2500.0 / 6.0, "h3f": None}      # h3f mixes 3-MFMA forward and 1-MFMA backward launches: its
#                                                              roofline is taken on executed MFMAs against the 16-bit peak
DENSE_16BIT_PEAK_TFLOPS = 2500.0
DTYPE_LABEL = {"bf16": "bf16", "fp16": "fp16", "f32x3": "fp32 storage, 3x bf16 MFMA per product",
               "f32h3": "fp32 storage, 3x MFMA per product (fp16 planes forward ~2^-21, bf16 planes backward)",
               "h3p": "fp16 plane pairs (hi + lo, 22 significant bits) split once by the producer, 3x fp16 MFMA per product "
                      "(~2^-21), fp32 accumulate; conv outputs / activation gradients fp32; loss-scaled backward",
               "f32x6": "fp32 storage, 6x bf16 MFMA per product (fp32-equivalent)",
               "h3f": "forward: fp16 plane pairs, 3x fp16 MFMA per product (bit-identical to h3p: logits, loss, BatchNorm "
                      "statistics); backward: 1 MFMA per product -- output gradients dy and the up half of d(concat) stored as "
                      "loss-scaled fp16, input-gradient weight planes fp16, weight gradients on the hi plane of the saved "
                      "plane-pair activations; conv outputs y and activation gradients da stay fp32 (ReLU masks and pool "
                      "positions decide as in h3p); fp32 accumulate, fp32 weights / optimiser"}
CONV_KERNELS = ("crimac_conv3x3: conv3x3_wch_kernel + conv3x3_p64_kernel + conv3x3_glds_w4_kernel + "
                "conv3x3_c16_kernel (halo-staged implicit-GEMM 3x3 conv, fwd + dgrad, all layers)")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=sorted(MFMA_PEAK_TFLOPS))
    ap.add_argument("--start-filts", type=int, default=64, help="128 = BASELINE configs[4] (2x channels)")
    ap.add_argument("--gpu-augment", action="store_true",
                    help="train on raw linear sv with add_noise / flip / dB on the GPU (configs[4])")
    ap.add_argument("--parity-precision", default="h3f", choices=["f32x3", "f32h3", "h3p", "h3f", "f32x6"],
                    help="h3f: h3p's forward pass (below) with the backward pass on the fp16 kernels -- the parity bar is on "
                         "the forward pass, gradients stay inside the reference's own fp32-vs-fp64 noise; "
                         "h3p: fp16 plane pairs split by the producer (fp32-class logits, identical argmax masks) at 3 MFMAs "
                         "per product on the LDS-DMA kernels; f32h3: the same arithmetic forward with fp32 storage (split "
                         "while staging) and a bf16-plane backward; f32x6: 6 MFMAs, fp32-equivalent gradients too")
    ap.add_argument("--roofline-steps", type=int, default=30, help="serialized steps of the per-kernel (roofline) pass")
    ap.add_argument("--roofline-warmup", type=int, default=10, help="serialized warm-up steps in front of that pass")
    ap.add_argument("--loop-iters", type=int, default=300, help="iterations of the train_loop leg (SegPipe.train_model + DataLoader)")
    ap.add_argument("--loop-workers", type=int, default=4,
                    help="DataLoader workers of the train_loop leg (the yaml's num_workers; 8 measured slower than 4 on the "
                         "16-core share of a GPU box: 2049 vs 2221 patches/s in bf16)")
    ap.add_argument("--no-train-loop", action="store_true")
    ap.add_argument("--tiled-pings", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true")
    ap.add_argument("--no-tiled", action="store_true")
    ap.add_argument("--tiled-ordered", action="store_true",
                    help="N > 1: after the timed tiled pass, one more pass with predict_survey(ordered_to_rank0=True) -- every chunk "
                         "handed to rank 0 point-to-point, asserted complete and in ping order (off by default: the timed legs of "
                         "the scaling run use collectives only; tools/runs/r5_03.sh rehearses it)")
    ap.add_argument("--no-wide", action="store_true", help="skip the configs[4] (wide net, fp16, GPU augmentation) leg")
    ap.add_argument("--spawn-selftest", action="store_true",
                    help="only start the ranks, all-reduce their ids and print the census (tests/test_bench_spawn.py)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------
# CPU baseline (oracle, "port")
# ----------------------------------------------------------------------------------------------------------
def cpu_baseline(parity_check=None):
    """Oracle train step and eval forward on the host cores, bounded to ~30 s (same synthetic
    crops, B = 2 and B = 32, median after warm-ups; the B = 32 train step is a single iteration).
    ``parity_check(x32, probs32) -> dict``: the oracle's B = 32 softmax output is handed to the caller, who compares the
    GPU path in the parity precision with it on the same 32 crops (the oracle as the checker: no extra CPU time)."""
    import torch
    from crimac_classifiers_unet_amd import synth
    from oracle import unet_oracle as orc
    # threads actually used: the affinity mask, capped at the GPU box's CPU share (16 per GPU) -- one thread per
    # VISIBLE core (256) on a cgroup-limited box stalls for minutes
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("CRIMAC_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline on {cores} threads (os.cpu_count()={os.cpu_count()})", file=sys.stderr, flush=True)
    sd = synth.synth_state_dict(seed=0)

    def data(batch):
        return (torch.from_numpy(synth.synth_echogram_batch(batch, 4, 256, 256, seed=1)),
                torch.from_numpy(synth.synth_labels(batch, 256, 256, seed=2)))

    def train_times(x, lab, warm, iters):
        state, vel, ts = sd, {}, []
        for i in range(warm + iters):
            t0 = time.perf_counter()
            _, _, grads, stats = orc.loss_and_grads(state, x, lab)
            state, vel = orc.sgd_momentum_step(dict(state), grads, vel, 0.005, 0.95)
            state.update(stats)
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return ts

    last = {}

    def infer_times(x, warm, iters):
        ts = []
        for i in range(warm + iters):
            t0 = time.perf_counter()
            last["logits"] = orc.predict(sd, x, return_softmax=False)
            torch.softmax(last["logits"], dim=1)
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return ts

    x2, l2 = data(2)
    t_train2 = statistics.median(train_times(x2, l2, 2, 5))
    t_inf2 = statistics.median(infer_times(x2, 2, 5))
    x32, l32 = data(32)
    t_inf32 = statistics.median(infer_times(x32, 1, 2))
    t_train32 = train_times(x32, l32, 1, 1)[0]
    parity32 = parity_check(x32, last["logits"]) if parity_check is not None else None
    return {"parity_batch32": parity32, "value": 2 / t_train2, "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": "oracle (torch CPU fp32) train step, batch 2 x 4x256x256, median of 5 after 2 warm-ups",
            "infer_value": 2 / t_inf2,
            "batch32": {"train_value": 32 / t_train32, "train_sample": "1 iteration after 1 warm-up",
                        "infer_value": 32 / t_inf32, "infer_sample": "median of 2 after 1 warm-up"}}


# ----------------------------------------------------------------------------------------------------------
# one precision mode: train step + inference + serialized per-kernel pass
# ----------------------------------------------------------------------------------------------------------
def measure_mode(args, precision, steps, warmup, world, rank, dev, grad_sync, infer=True, log=print, serial_pass=True):
    import torch
    import torch.distributed as dist
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import hip, synth

    sf = args.start_filts
    model = pkg.UNet_Baseline(3, 4, start_filts=sf, precision=precision)
    model.load_state_dict(synth.synth_state_dict(start_filts=sf, seed=0))      # random-init weights of the architecture
    model.to(dev).train()
    eng = model.engine
    B = args.batch
    lab = torch.from_numpy(synth.synth_labels(B, 256, 256, seed=200 + rank)).to(dev)
    cw = torch.tensor([10.0, 300.0, 250.0], device=dev)
    x = torch.from_numpy(synth.synth_echogram_batch(B, 4, 256, 256, seed=100 + rank)).to(dev)
    if args.gpu_augment:
        x_lin = torch.pow(10.0, x / 10.0)             # raw linear sv crops, as the reader hands them over
        it = [0]

        def step():
            it[0] += 1
            return eng.train_step_augmented(x_lin, lab, cw, 0.005, 0.95, seed=(rank << 32) ^ it[0],
                                            grad_sync=grad_sync)
    else:
        def step():
            return eng.train_step(x, lab, cw, lr=0.005, momentum=0.95, grad_sync=grad_sync)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(t):
        if world > 1:
            tt = torch.tensor([t], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt)
        return t

    log(f"{precision}: model ready on {dev} (start_filts {sf}), batch {B}/GPU, world {world}")
    for _ in range(warmup):
        loss = step()
    barrier()
    # the cyclic garbage collector stays out of the timed region (the instrumented steps create ~250 event objects each;
    # a full collection of a process that has run other legs before costs tens of ms -- more than a step)
    import gc
    gc.collect()
    gc.disable()
    # The timed region runs WITHOUT per-launch instrumentation: two HIP events around each of ~60 launches per step cost the
    # host 0.4-0.6 ms per step on a fresh process (bf16 11.2 vs 11.6 ms, h3f 17.7 vs 18.2) and 1.5-2 ms in a process that
    # has created and destroyed tens of thousands of events in earlier legs (the parity leg of the default run read 19.6 ms
    # where the same binaries alone read 18.2).  The overlapped per-launch durations come from a short instrumented pass
    # right behind the timed region.
    if world > 1:
        eng.exchange_probe = []      # (event after the backward pass, event after the last collective) per step
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    n_ovl = max(3, min(steps, 6))
    hip.PROFILE = []                 # HIP events around the conv / wgrad launches, side stream ON (overlapped durations)
    for _ in range(n_ovl):
        step()
    barrier()
    gc.enable()
    prof_timed, hip.PROFILE = hip.PROFILE, None
    exchange = None
    if world > 1:
        ex = [a.elapsed_time(b) for a, b in eng.exchange_probe]
        eng.exchange_probe = None
        exchange = {"exchange_ms_exposed": max_over_ranks(statistics.median(ex)) if ex else None,
                    "exchange_ms_exposed_worst_step": max_over_ranks(max(ex)) if ex else None,
                    "what": "per step, time on the training stream between the end of the backward pass and the moment the "
                            "last gradient collective has been waited for (the part of the 124 MB exchange the backward "
                            "pass did not hide); median over the timed steps, max over ranks",
                    "algo": getattr(grad_sync, "algo", None), "bucket_mb": grad_sync.bucket_elems * 4 / (1 << 20),
                    "per_range_optimiser_step": bool(eng.early_sgd and eng.early_sgd_multi)}
    final_loss = float(loss)
    assert final_loss == final_loss, "training diverged (NaN loss)"
    skipped = eng.skipped_steps()
    assert eng.loss_scale == 1.0 or skipped < steps + warmup, (
        f"every loss-scaled step was skipped (scale {eng.loss_scale}): the number would be the throughput of steps "
        "whose update never ran")
    log(f"{precision}: timed region done: {1e3 * elapsed / steps:.2f} ms/step")

    # Per-kernel durations: in the timed region the weight gradients run on a side stream CONCURRENTLY with the
    # input-gradient convolutions, so their HIP-event intervals overlap.  The roofline figures come from a second,
    # serialized pass (same steps, side stream off) right after the timed region: >= 30 steps after >= 10 warm-ups,
    # every launch of the step reduced to the MEDIAN (and minimum) over the steps, and an MFMA-only calibration launch
    # (crimac_mfma_calibrate: rate + in-kernel shader clock) before and after it -- a pass taken on a throttled box
    # shows in the line itself.
    prof, serialized, n_ser, calib = prof_timed, False, n_ovl, None
    if eng.wgrad_side_streams > 0 and serial_pass:
        saved_cfg = (eng.wgrad_side_streams, eng._side)
        eng.wgrad_side_streams, eng._side = 0, None
        for _ in range(args.roofline_warmup):
            step()
        barrier()
        calib = {"before": mfma_calibration(dev)}
        n_ser = max(3, args.roofline_steps)
        hip.PROFILE = []
        for _ in range(n_ser):
            step()
        barrier()
        prof, hip.PROFILE = hip.PROFILE, None
        calib["after"] = mfma_calibration(dev)
        eng.wgrad_side_streams, eng._side = saved_cfg
        serialized = True

    def per_launch(records, n_steps):
        """[(name, flops, median ms, min ms, mean ms)] per launch POSITION of the step, over the n_steps repetitions."""
        n = len(records) // max(n_steps, 1)
        if n == 0 or n * n_steps != len(records):
            raise RuntimeError(f"profile records ({len(records)}) are not a multiple of the steps ({n_steps})")
        out = []
        for p in range(n):
            ts = [records[k * n + p][2].elapsed_time(records[k * n + p][3]) for k in range(n_steps)]
            assert all(records[k * n + p][0] == records[p][0] for k in range(n_steps))
            out.append((records[p][0], records[p][1], statistics.median(ts), min(ts), sum(ts) / len(ts),
                        records[p][4] if len(records[p]) > 4 else 1))
        return out

    pl_ser, pl_timed = per_launch(prof, n_ser), per_launch(prof_timed, n_ovl)
    peak = MFMA_PEAK_TFLOPS[precision]
    mixed = peak is None              # (h3f) launches of different MFMAs-per-product in one step
    scale_f = (sf / 64.0) ** 2        # conv FLOPs scale with the square of the width (first / last layer aside)

    def roofline(kname, label):
        sel = [r for r in pl_ser if r[0].startswith(kname)]            # (crimac_wgrad[_partials])
        sel_t = [r for r in pl_timed if r[0].startswith(kname)]
        n = max(len(sel), 1)
        fl = sum(r[1] for r in sel)
        med, mn, mean = (sum(r[i] for r in sel) for i in (2, 3, 4))
        if mixed:
            # executed MFMA FLOPs (algorithmic FLOPs x MFMAs per product of each launch) against the dense 16-bit peak:
            # for a pure mode this is the same fraction as algorithmic FLOPs against peak / MFMAs-per-product
            fl_x = sum(r[1] * r[5] for r in sel)
            ach = fl_x / (med * 1e-3) / 1e12 if med > 0 else 0.0
            pk = DENSE_16BIT_PEAK_TFLOPS
        else:
            fl_x = fl
            ach = fl / (med * 1e-3) / 1e12 if med > 0 else 0.0
            pk = peak
        r = {"bound": "mfma", "kernel": label, "achieved": ach, "peak": pk, "unit": "TFLOP/s",
             "frac": ach / pk, "traffic": None,
             "algorithmic_flops_per_launch": fl / n, "launches_per_step": len(sel), "steps_measured": n_ser,
             "median_launch_us": 1e3 * med / n, "min_launch_us": 1e3 * mn / n, "avg_launch_us": 1e3 * mean / n,
             "frac_from_min": (fl_x / (mn * 1e-3) / 1e12 / pk) if mn > 0 else 0.0,
             "frac_from_mean": (fl_x / (mean * 1e-3) / 1e12 / pk) if mean > 0 else 0.0,
             "measured": ((f"serialized pass after the timed region (weight-gradient side stream off), {n_ser} steps after "
                           f"{args.roofline_warmup} warm-ups; achieved = algorithmic FLOPs of one step's launches / the sum of "
                           "their per-launch MEDIAN HIP-event durations over the steps (avg_launch_us = the mean, what "
                           "rocprofv3 --stats averages); in the timed region these launches overlap the weight gradients")
                          if serialized else "timed region"),
             "median_launch_us_overlapped": 1e3 * sum(r[2] for r in sel_t) / max(len(sel_t), 1),
             "overlapped_measured": f"{n_ovl} instrumented steps right behind the (uninstrumented) timed region, side stream on"}
        if mixed:
            r["achieved_is"] = ("EXECUTED MFMA TFLOP/s: algorithmic FLOPs of each launch x its MFMAs per product (3 for the "
                                "plane-pair forward launches, 1 for the fp16 backward launches); algorithmic rate: "
                                f"{fl / (med * 1e-3) / 1e12 if med > 0 else 0.0:.1f} TFLOP/s")
        if calib is not None:
            r["mfma_calibration"] = calib
        return r

    res = {"precision": precision, "dtype": DTYPE_LABEL[precision],
           "train_patches_per_s": world * B * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
           "train_tflops": world * B * steps / elapsed * TRAIN_GFLOP_PER_PATCH * scale_f / 1e3,
           "final_loss": final_loss, "loss_scale": eng.loss_scale, "skipped_steps": skipped, "exchange": exchange,
           "roofline": roofline("crimac_conv3x3", CONV_KERNELS if precision in ("bf16", "fp16") else
                                ("crimac_conv3x3: conv3x3_wch_kernel (plane-pair forms: 3 MFMAs per fragment pair) + conv3x3_c16_kernel "
                                 "(first layer, pseudo-channels)" if precision == "h3p" else
                                 "crimac_conv3x3: forward = the plane-pair forms of conv3x3_wch_kernel / conv3x3_c16_kernel, input "
                                 "gradients = the fp16 forms of conv3x3_wch_kernel / conv3x3_p64_kernel" if precision == "h3f" else
                                 "crimac_conv3x3: conv3x3_kernel (fp32 storage, split-bf16 planes, register-staged halo)")),
           "roofline_wgrad": roofline("crimac_wgrad", "wgrad_pp_kernel + wgrad_up_pp_kernel + wgrad_kernel (weight gradient, all shapes)"
                                      if precision == "h3p" else
                                      "wgrad_group_kernel (conv3x3 layers of a gradient range in one launch) + wgrad_kernel (first "
                                      "layer, transposed convolutions)")}
    if infer:
        model.eval()
        with torch.no_grad():
            for _ in range(max(warmup // 2, 2)):
                model.predict_softmax(x)
            barrier()
            t1 = time.perf_counter()
            for _ in range(steps):
                model.predict_softmax(x)
            barrier()
            ti = max_over_ranks(time.perf_counter() - t1)
        res["infer_patches_per_s"] = world * B * steps / ti
        res["infer_tflops"] = res["infer_patches_per_s"] * FWD_GFLOP_PER_PATCH * scale_f / 1e3
    return res, model


_CALIB = {}


def mfma_calibration(dev, iters=15000, blocks=512):
    """One MFMA-only launch (crimac_mfma_calibrate, ~2 ms: 512 workgroups x 4 waves x iters x 8 MFMAs 16x16x32 bf16 on
    register operands) timed with HIP events, median of 3 after one warm-up: the dense-MFMA rate the chip delivers at
    this moment and the shader clock its workgroups saw (s_memtime / s_memrealtime, median over the workgroups)."""
    import numpy as np
    import torch
    from crimac_classifiers_unet_amd import hip
    if "stamps" not in _CALIB:
        _CALIB["stamps"] = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
        _CALIB["sink"] = torch.zeros(4, dtype=torch.float32, device=dev)
    st, sink = _CALIB["stamps"], _CALIB["sink"]
    ts = []
    for k in range(4):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        hip.call("crimac_mfma_calibrate", iters, blocks, hip.ptr(st), hip.ptr(sink))
        e_.record()
        e_.synchronize()
        if k:
            ts.append(s_.elapsed_time(e_))
    ms = statistics.median(ts)
    v = st.cpu().numpy().reshape(blocks, 2).astype(np.float64)
    ghz = float(np.median(v[:, 0] / np.maximum(v[:, 1], 1.0)) * 0.1)
    flop = 2.0 * 16 * 16 * 32 * 8 * iters * 4 * blocks
    tf = flop / (ms * 1e-3) / 1e12
    return {"tflops": tf, "frac_of_nominal_peak": tf / 2500.0, "shader_clock_ghz": ghz, "launch_ms": ms,
            "what": "MFMA-only launch (register operands, no memory traffic), observed in this run"}


def golden_parity(precision, dev, log):
    """Parity of the binaries this run times, measured here and now: the 2 x 4 x 256 x 256 golden crop of the imported
    reference (tests/golden/full64_256.npz, tools/make_golden.py: reference UNet_Baseline on the CPU, same synthetic
    weights and crops) through the engine in ``precision`` -- eval logits (BatchNorm running statistics), train-mode
    logits (batch statistics) and the weighted cross entropy.  rel = max |delta| / max |reference| (the north-star
    bar is 1e-3 with identical argmax masks)."""
    import numpy as np
    import torch
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import synth
    path = os.path.join(ROOT, "tests", "golden", "full64_256.npz")
    if not os.path.exists(path):
        return {"error": "tests/golden/full64_256.npz not found"}
    fix = np.load(path)
    model = pkg.UNet_Baseline(3, 4, precision=precision)
    model.load_state_dict(synth.synth_state_dict(seed=0))
    model.to(dev)
    x = torch.from_numpy(synth.synth_echogram_batch(2, 4, 256, 256, seed=1)).to(dev)
    lab = torch.from_numpy(synth.synth_labels(2, 256, 256, seed=2)).to(dev)

    def cmp(out, ref):
        out, ref = out.detach().float().cpu(), torch.from_numpy(ref)
        return (float((out - ref).abs().max() / ref.abs().max()), int((out.argmax(1) != ref.argmax(1)).sum()))

    model.eval()
    with torch.no_grad():
        r_e, f_e = cmp(model(x), fix["logits_eval"])
    model.train()
    crit = pkg.WeightedCrossEntropy([10.0, 300.0, 250.0]).to(dev)
    logits = model(x)
    loss = float(crit(logits, lab.long()))
    r_t, f_t = cmp(logits, fix["logits_train"])
    ref_loss = float(fix["losses"][0])
    res = {"crop": "tests/golden/full64_256.npz (reference UNet_Baseline on CPU, 2 x 4 x 256 x 256)",
           "pixels": int(fix["logits_eval"][:, 0].size),
           "eval_logits_rel": r_e, "eval_argmax_flips": f_e, "train_logits_rel": r_t, "train_argmax_flips": f_t,
           "loss_rel": abs(loss - ref_loss) / abs(ref_loss),
           "meets_north_star": bool(r_e <= 1e-3 and f_e == 0)}
    log(f"{precision}: golden crop: eval rel {r_e:.2e} flips {f_e}, train rel {r_t:.2e} flips {f_t}, "
        f"loss rel {res['loss_rel']:.2e}")
    del model
    return res


def measure_tiled(model, args, log, world=1):
    """BASELINE configs[3]: whole-survey tiled inference, end to end (host reader included).  world > 1: the chunks of
    the survey are sharded over the ranks (rank r owns chunks r, r + N, ...; no collective in the data path); the
    figure is all patches of the survey / the slowest rank's time."""
    import types
    import torch
    import torch.distributed as dist
    from crimac_classifiers_unet_amd import synth, tiled_inference as ti
    n_pings, n_range, preload = args.tiled_pings, 1024, 4096
    t0 = time.perf_counter()
    reader = synth.SyntheticSurveyReader(n_pings=n_pings, n_range=n_range, seabed_index=900, block=preload)
    log(f"tiled: synthetic survey 4 x {n_pings} x {n_range} built in {time.perf_counter() - t0:.1f} s")
    pipe = types.SimpleNamespace(model=model, device=next(model.parameters()).device, frequencies=[18, 38, 120, 200])
    import numpy as np
    # float16: what the reference stores (save_predict.py:212); N > 1: every rank keeps its own ping ranges (the
    # zero-communication form; predict_survey's default hands the chunks to rank 0 in ping order for a sequential writer)
    f16 = dict(out_dtype=np.float16, ordered_to_rank0=False)
    for _ in ti.predict_survey(reader, pipe, (256, 256), 20, args.batch, preload, **f16):
        pass                                             # warm-up: one untimed pass over the survey
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    n_patches, written, n_mine = 0, 0, 0
    for s, e, out in ti.predict_survey(reader, pipe, (256, 256), 20, args.batch, preload, **f16):
        n_patches += len(ti.plan_grid(n_range, 900, s, e))
        written += int((out[0, :, ::64] != 0).sum())
        n_mine += e - s
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dev = next(model.parameters()).device
        agg = torch.tensor([n_patches, written, n_mine], dtype=torch.float64, device=dev)
        dist.all_reduce(agg)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        n_patches, written, n_mine, dt = int(agg[0]), int(agg[1]), int(agg[2]), float(tmax)
        assert n_mine == n_pings, (n_mine, n_pings)          # the ranks covered the survey exactly once
    handoff = None
    if world > 1 and args.tiled_ordered:
        # the opt-in ordered hand-off (predict_survey(ordered_to_rank0=True)): rank 0 must receive EVERY chunk of the survey
        # in ping order (what the reference's sequential append_to_zarr writer needs), the other ranks yield nothing
        dist.barrier()
        t1 = time.perf_counter()
        got = [(s, e, int((out[0, :, ::64] != 0).sum())) for s, e, out in
               ti.predict_survey(reader, pipe, (256, 256), 20, args.batch, preload, out_dtype=np.float16, ordered_to_rank0=True)]
        torch.cuda.synchronize()
        dt_o = time.perf_counter() - t1
        dev = next(model.parameters()).device
        cnt = torch.tensor([len(got)], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt)
        tmax_o = torch.tensor([dt_o], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax_o, op=dist.ReduceOp.MAX)
        if dist.get_rank() == 0:
            plan = ti.plan_chunks(0, n_pings, preload)
            assert [g[:2] for g in got] == plan, "ordered hand-off: rank 0 did not get every chunk in ping order"
            assert int(cnt) == len(plan), "ordered hand-off: a rank other than 0 yielded chunks"
            handoff = {"chunks_on_rank0": len(got), "chunks_of_survey": len(plan), "in_ping_order": True,
                       "written_sampled": sum(g[2] for g in got), "seconds": float(tmax_o),
                       "patches_per_s": n_patches / float(tmax_o),
                       "what": "predict_survey(ordered_to_rank0=True): chunk-sharded compute, every finished float16 chunk handed "
                               "to rank 0 point-to-point (isend / irecv), rank 0 yields the survey in ping order"}
        else:
            assert got == [], "ordered hand-off: a rank other than 0 yielded chunks"
    ti.release_staging()          # (0.5 GB of pinned host memory + two device chunk buffers kept between surveys: freed before
    #                                the next leg is timed -- the parity leg ran 6 % slower behind a tiled leg that kept them)
    return {"workload": f"BASELINE configs[3]: synthetic survey sv [4, {n_pings}, {n_range}] fp32, flat seabed 900, "
                        f"preload_n_pings {preload}, patch 256, overlap 20, "
                        + ("1 GPU streamed" if world == 1 else f"chunks sharded over {world} ranks, no data-path collective")
                        + f"; the {len(ti.plan_grid(n_range, 900, 0, preload))} "
                        f"patches of a chunk run as forward calls of up to {max(args.batch, ti.INTERNAL_BATCH)} patches "
                        f"(eval mode: results do not depend on the batch; the caller's batch_size {args.batch} is a lower bound)",
            "patches_per_forward_call": max(args.batch, ti.INTERNAL_BATCH),
            "patches_per_s": n_patches / dt, "pings_per_s": n_pings / dt, "n_patches": n_patches, "seconds": dt,
            "precision": model.precision,
            "timed": "host reader + H2D + crop/dB gather + U-Net + softmax + scatter + D2H of [2, range, pings] float16",
            "ordered_handoff": handoff,
            "written_frac_sampled": written / (n_range * ((n_pings + 63) // 64))}


_LOOP_DL = {}
LOOP_MODES = {"transformed": 0, "raw_f32": 1, "raw_f64": 2, "host_chain": 3}


class _HostChain:
    """The reference's worker-side transform chain (batch/dataset.py:89-103) on one raw crop -- the CPU-baseline leg of
    ``train_loop_raw``: oracle/augment_oracle.worker_train_chain (pinned bit for bit against the imported reference,
    tests/test_worker_chain.py), with a RandomState per sample."""

    def __call__(self, data, labels, index):
        import numpy as np
        from oracle import augment_oracle           # (CPU baseline leg only)
        return augment_oracle.worker_train_chain(data, labels, np.random.RandomState(0x5EED + index))


class _LegSampler:
    """The index range of the leg being measured; lives in the parent, so one set of worker processes serves all legs."""
    lo = hi = 0

    def __iter__(self):
        return iter(range(self.lo, self.hi))

    def __len__(self):
        return self.hi - self.lo


def train_loop_dataloader(args):
    """ONE DataLoader (persistent workers) for all train_loop legs: forking a worker out of this process -- tens of GB of
    GPU mappings by now -- takes seconds, and six legs would fork twenty-four.  The leg is encoded in the sample index
    (index >> 32, LOOP_MODES); the sampler (parent side) hands out the index range of the current leg."""
    import numpy as np
    import torch
    from crimac_classifiers_unet_amd import synth
    if "dl" in _LOOP_DL:
        return _LOOP_DL["dl"], _LOOP_DL["sampler"]
    B = args.batch
    n_distinct = 2 * B
    data = synth.synth_echogram_batch(n_distinct, 4, 256, 256, seed=300)
    labels = synth.synth_labels(n_distinct, 256, 256, seed=301)
    # raw legs: an in-memory survey with annotated schools (the label refinement has its real work to do), NaN / Inf samples
    survey = synth.SyntheticSurveyReader(n_pings=16384, n_range=1024, block=4096, schools=60, bad_frac=1e-4, seed=5)
    raw = {1: synth.RawCropDataset(survey, (256, 256), 1 << 31, seed=21, dtype=np.float32),
           2: synth.RawCropDataset(survey, (256, 256), 1 << 31, seed=21, dtype=np.float64),
           3: synth.RawCropDataset(survey, (256, 256), 1 << 31, seed=21, dtype=np.float32, transform=_HostChain())}

    class LoopCrops(torch.utils.data.Dataset):
        def __len__(self):
            return 1 << 34

        def __getitem__(self, index):
            mode, i = index >> 32, index & 0xFFFFFFFF
            if mode == 0:        # pre-transformed crops at zero host cost: the hand-over / collate / H2D machinery alone
                k = i % n_distinct
                return {"data": data[k], "labels": labels[k], "center_coordinates": np.array([128, 128 + i], dtype=np.int64)}
            return raw[mode][i]

    nw = max(int(args.loop_workers), 0)
    sampler = _LegSampler()
    _LOOP_DL["sampler"] = sampler
    _LOOP_DL["dl"] = torch.utils.data.DataLoader(LoopCrops(), batch_size=B, sampler=sampler, num_workers=nw, drop_last=True,
                                                 persistent_workers=nw > 0)
    return _LOOP_DL["dl"], sampler


def measure_train_loop(args, precision, dev, log, resident_patches_per_s, pin_batches=True, mode="transformed", iters=None):
    """The loop the reference actually runs (pipeline.py:161-181): ``SegPipeUNet.train_model`` fed by a
    ``torch.utils.data.DataLoader`` (default collate, worker processes) over an in-memory synthetic Dataset that yields
    the reference's batch dict -- ``data`` [4, 256, 256], ``labels`` int16 [256, 256], ``center_coordinates``
    int64 [2] per sample -- so the timed region contains the collate, the worker -> parent hand-over,
    the H2D copy (pinned ring + copy stream, staging.py) and the step.  One untimed pass over the DataLoader first
    (worker start-up, pinned / device allocations), then one timed pass; workers persist between the two.

    ``mode`` (LOOP_MODES):
      transformed  the Dataset returns ready-made dB crops and final labels at zero host cost (round 4's ``train_loop``):
                   the staging machinery alone;
      raw_f32 /    ``train_loop_raw``: every ``__getitem__`` draws a random centre and GATHERS a raw crop from an in-memory
      raw_f64      survey (linear sv + raw annotation ids: the reference's Dataset built with its three transform hooks set
                   to None, batch/dataset.py:76-110; float32 = the memmap flavour's crop dtype, float64 = the zarr
                   flavour's), and ``SegPipeUNet(gpu_augment=True, gpu_label_transform=True)`` runs add_noise / flip /
                   refine_label_boundary / convert_label_indexing / remove_nan_inf / db_with_limits on the GPU in front of
                   the step -- all inside the timed region;
      host_chain   the CPU baseline of that leg: the same crops with the reference's transform chain run IN THE WORKERS
                   (oracle/augment_oracle.worker_train_chain == the imported reference bit for bit), default pipeline."""
    import numpy as np
    import torch
    import crimac_classifiers_unet_amd as pkg
    from crimac_classifiers_unet_amd import synth

    B = args.batch
    iters = int(iters) if iters else max(args.loop_iters, 30)     # (a pass pays the DataLoader's start-up once)
    nw = max(int(args.loop_workers), 0)
    dl, sampler = train_loop_dataloader(args)
    sampler.lo = LOOP_MODES[mode] << 32
    sampler.hi = sampler.lo + iters * B
    on_gpu = mode in ("raw_f32", "raw_f64")
    pipe = pkg.SegPipeUNet(checkpoint_dir=None, data_mode="zarr", frequencies=[18, 38, 120, 200], patch_size=[256, 256],
                           loss_type="CE", lr=0.005, lr_reduction=0.5, lr_step=1000, momentum=0.95, batch_size=B,
                           num_workers=nw, iterations=iters, test_iter=10, log_step=10 ** 9, save_model_params=False,
                           meta_channels=[], late_meta_inject=False, eval_mode="all", experiment_name="bench",
                           precision=precision, infer_precision=precision, pin_batches=pin_batches,
                           loss_flush=10 ** 9, gpu_augment=on_gpu, gpu_label_transform=on_gpu)
    pipe.model.load_state_dict(synth.synth_state_dict(seed=0))

    class LastLoss:
        last = None

        def add_scalar(self, tag, scalar_value, global_step):
            if tag == "train/loss":
                self.last = scalar_value

    import contextlib
    lg = LastLoss()
    phases = {}
    if pin_batches:
        pipe.stager_stats = phases                    # seconds per staging phase (staging.BatchStager._note)
    with contextlib.redirect_stdout(sys.stderr):      # (train_model prints "Training complete" like the reference)
        pipe.train_model(dl, None, logger=lg)         # untimed: workers, allocations, first-touch
        torch.cuda.synchronize()
        phases.clear()
        t0 = time.perf_counter()
        pipe.train_model(dl, None, logger=lg)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    del dl, pipe                                      # (the model goes before anything else is timed; the workers stay for the next leg)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    pps = iters * B / dt
    assert lg.last is not None and lg.last == lg.last, "train_loop: NaN loss"
    log(f"{precision}: train_loop[{mode}] ({'pinned ring' if pin_batches else 'in-line copy'}, {nw} workers): {pps:.0f} patches/s "
        f"= {pps / resident_patches_per_s:.3f} of the resident-batch figure")
    return {"patches_per_s": pps, "ms_per_step": 1e3 * dt / iters, "iterations": iters, "batch": B,
            "dataset": mode, "dataloader_workers": nw, "input_staging": "pinned ring + copy stream (staging.BatchStager)" if pin_batches
            else "in-line .to(device) as the reference (pipeline.py:163-164)",
            "vs_resident_batch": pps / resident_patches_per_s, "final_loss": lg.last,
            "host_phases_ms_per_step": {k[:-2]: round(1e3 * v / iters, 3) for k, v in sorted(phases.items())} or None,
            "timed": {"transformed": "SegPipeUNet.train_model over a DataLoader (default collate, batch dict of the reference): "
                                     "collate + hand-over + H2D + step, one full pass of the DataLoader after one untimed pass",
                      "raw_f32": "random centre + crop gather from the in-memory survey in the workers (float32 crops), collate, "
                                 "hand-over, H2D, add_noise / flip / label refinement + indexing / remove_nan_inf / dB ON THE GPU, step",
                      "raw_f64": "as raw_f32 with float64 crops (what the reference's zarr crop returns); the `.float()` of "
                                 "pipeline.py:163 is taken per crop in the workers' collate (staging.collate_float32, yaml key "
                                 "collate_float32), so collate, hand-over and H2D move float32",
                      "host_chain": "random centre + crop gather + the reference's transform chain (add_noise, flip, label "
                                    "refinement + indexing, remove_nan_inf, dB) IN THE WORKERS, collate, hand-over, H2D, step"}[mode]}


def load_profile_json(name):
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        d["_file"] = "profiles/" + name
        return d
    except Exception:
        return None


# ----------------------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist
    from crimac_classifiers_unet_amd import parallel

    world, rank, local = parallel.init_distributed()
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; measuring WORLD_SIZE ranks", file=sys.stderr)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    if args.spawn_selftest:
        # census of the ranks that really started (CPU-testable: gloo when there is no GPU)
        t = torch.ones(1)
        if torch.cuda.is_available():
            t = t.cuda(local % max(torch.cuda.device_count(), 1))
        gs_ok = None
        if world > 1:
            dist.all_reduce(t)
            # the gradient exchange itself at this world size: four ranges handed over in backward order, 32 MB-style
            # buckets scaled down, all-reduce and reduce-scatter + all-gather spellings, per-range completion
            gs_ok = True
            n = 8 * 4096 + 64
            bounds = [(n - 9000, n), (n - 20000, n - 9000), (4000, n - 20000), (0, 4000)]
            expect = torch.arange(n, dtype=torch.float32) * (world * (world + 1) // 2)
            for algo in ("all_reduce", "rs_ag"):
                flat = (torch.arange(n, dtype=torch.float32) * (rank + 1)).to(t.device)
                gs = parallel.GradSync(bucket_mb=4096 * 4 / (1 << 20), algo=algo)
                for lo, hi in bounds:
                    gs.launch(flat, lo, hi)
                for lo, hi in bounds:
                    gs.finish_range(lo, hi)
                    gs_ok = gs_ok and bool(torch.equal(flat[lo:hi].cpu(), expect[lo:hi]))
                gs_ok = gs_ok and abs(gs.finish() - 1.0 / world) < 1e-12
            ok_t = torch.tensor([1.0 if gs_ok else 0.0]).to(t.device)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)          # (every rank saw the right sums)
            gs_ok = bool(ok_t.item() == 1.0)
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "ranks_counted": int(t.item()),
                              "backend": dist.get_backend() if world > 1 else None, "gradsync_ok": gs_ok,
                              "self_launched": os.environ.get("CRIMAC_SELF_LAUNCHED") == "1"}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    local = local % max(torch.cuda.device_count(), 1)     # (rehearsals with several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    census = None
    if world > 1:
        # which ranks really take part, on which devices: an all-reduce of ones and an all-gather of the device indices
        # over the SAME backend the gradients use
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        devs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(devs, torch.tensor([local], dtype=torch.int64, device=dev))
        census = {"ranks_counted": int(ones.item()), "backend": dist.get_backend(),
                  "device_of_rank": [int(d.item()) for d in devs],
                  "distinct_devices": len({int(d.item()) for d in devs})}
    grad_sync = parallel.GradSync()
    main, model = measure_mode(args, args.precision, args.steps, args.warmup, world, rank, dev, grad_sync,
                               infer=not args.no_infer, log=log)
    tiled = None
    if not args.no_tiled and args.start_filts == 64:
        tiled = measure_tiled(model, args, log, world)
        log(f"tiled: {tiled['patches_per_s']:.0f} patches/s end to end")
    del model
    parity = None
    if world == 1 and not args.no_parity_mode and args.precision != args.parity_precision and args.start_filts == 64:
        torch.cuda.empty_cache()
        parity, pm = measure_mode(args, args.parity_precision, max(3, args.steps), max(2, args.warmup), world, rank, dev,
                                  grad_sync, infer=not args.no_infer, log=log)
        if not args.no_tiled:
            parity["tiled"] = measure_tiled(pm, args, log)       # configs[3] in the parity precision as well
            log(f"tiled ({args.parity_precision}): {parity['tiled']['patches_per_s']:.0f} patches/s end to end")
        del pm
        parity["golden_parity"] = golden_parity(args.parity_precision, dev, log)
        if args.parity_precision == "h3f":
            # the all-plane-pair variant (backward pass on plane pairs too: 3 MFMAs per product everywhere), timed region only
            torch.cuda.empty_cache()
            hp, hm = measure_mode(args, "h3p", max(3, args.steps // 2), max(2, args.warmup), world, rank, dev, grad_sync,
                                  infer=False, log=log, serial_pass=False)
            del hm
            parity["h3p_plane_pair_backward"] = {"train_patches_per_s": hp["train_patches_per_s"], "ms_per_step": hp["ms_per_step"],
                                                 "what": "precision 'h3p': the same forward pass, backward pass on fp16 plane pairs "
                                                         "(the parity mode of round 3)"}
    main["golden_parity"] = golden_parity(args.precision, dev, log) if (args.start_filts == 64 and rank == 0) else None

    wide = None
    if world == 1 and not args.no_wide and args.start_filts == 64 and not args.gpu_augment:
        # BASELINE configs[4] on one GPU, driver-timed: 2x channels, fp16 + MFMA with loss scaling, on-GPU add_noise / flip
        torch.cuda.empty_cache()
        wargs = argparse.Namespace(**vars(args))
        wargs.start_filts, wargs.gpu_augment = 128, True
        wres, wm = measure_mode(wargs, "fp16", 3, 2, world, rank, dev, grad_sync, infer=not args.no_infer, log=log)
        del wm
        torch.cuda.empty_cache()
        wide = {"workload": "BASELINE configs[4] on ONE GPU: wide U-Net (depth 5, 128 filters), batch "
                            f"{args.batch} x 4x256x256, fp16 + MFMA, dynamic loss scaling, on-GPU add_noise / flip / dB",
                "train_patches_per_s": wres["train_patches_per_s"], "ms_per_step": wres["ms_per_step"],
                "infer_patches_per_s": wres.get("infer_patches_per_s"), "train_tflops": wres["train_tflops"],
                "skipped_steps": wres["skipped_steps"], "loss_scale": wres["loss_scale"],
                "conv_frac_of_peak": wres["roofline"]["frac"], "wgrad_frac_of_peak": wres["roofline_wgrad"]["frac"],
                "steps": 3, "warmup": 2}
        log(f"wide fp16 (configs[4], 1 GPU): {wide['train_patches_per_s']:.0f} patches/s")
    # The loop the reference runs (DataLoader + train_model), LAST of the GPU legs: its DataLoader forks worker processes
    # from this process, and the legs measured after it in the first version of this bench ran 6 % slower
    # (h3p step 27.2 ms in the default run against 25.6 ms on its own, same box, same per-kernel durations)
    if world == 1 and not args.no_train_loop and args.start_filts == 64 and not args.gpu_augment:
        torch.cuda.empty_cache()
        main["train_loop"] = measure_train_loop(args, args.precision, dev, log, main["train_patches_per_s"])
        main["train_loop"]["inline_copy"] = measure_train_loop(args, args.precision, dev, log, main["train_patches_per_s"],
                                                               pin_batches=False)
        if parity is not None:
            parity["train_loop"] = measure_train_loop(args, args.parity_precision, dev, log, parity["train_patches_per_s"])
        # train_loop_raw: the loop a drop-in user runs -- real crop gathers in the workers, raw crops up,
        # the whole transform chain on the GPU -- next to its CPU baseline (the same chain in the workers)
        raw = measure_train_loop(args, args.precision, dev, log, main["train_patches_per_s"], mode="raw_f32")
        raw["float64_crops"] = measure_train_loop(args, args.precision, dev, log, main["train_patches_per_s"], mode="raw_f64")
        hc = measure_train_loop(args, args.precision, dev, log, main["train_patches_per_s"], mode="host_chain",
                                iters=max(12, args.loop_iters // 8))
        nw_ = max(int(args.loop_workers), 1)
        hc["host_ms_per_patch_per_worker"] = 1e3 * nw_ / hc["patches_per_s"]
        hc["what"] = ("CPU baseline of train_loop_raw, kind 'port': the reference's per-sample transform chain in the DataLoader "
                      "workers (reference measured in the build container: 12.4 ms per float32 patch and core -- add_noise 8.3, "
                      "label transform 3.0, dB 0.9; profiles/r05_reference_host_chain.txt)")
        raw["host_chain_baseline"] = hc
        raw["speedup_vs_host_chain"] = raw["patches_per_s"] / hc["patches_per_s"]
        main["train_loop_raw"] = raw
        if parity is not None:
            parity["train_loop_raw"] = measure_train_loop(args, args.parity_precision, dev, log, parity["train_patches_per_s"],
                                                          mode="raw_f32")
        _LOOP_DL.clear()                                   # (the worker processes end here, before the CPU baseline is timed)
        import gc
        gc.collect()
    if rank == 0:
        sf = args.start_filts
        rl = main["roofline"]
        def replay(roof, roof_w, prec):
            """HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes) and MFMA busy fraction (SQ_VALU_MFMA_BUSY_CYCLES /
            GRBM_GUI_ACTIVE pass) of the same command, from the committed profiles of this round -- labelled as replayed."""
            pm = load_profile_json(f"r05_{prec}_pmc_traffic.json") or load_profile_json(f"r04_{prec}_pmc_traffic.json")
            if pm:
                for r_, pat in ((roof, "conv3x3"), (roof_w, "wgrad")):
                    sel = [v for k, v in pm["kernels"].items() if k.startswith(pat)]
                    nl = sum(v["launches"] for v in sel)
                    # (the mean over all launches of the class, like `achieved`; for the weight gradients that mixes
                    # the grouped launches -- 1.5 GB each in bf16 -- with the single-layer ones, 0.3 GB)
                    r_["traffic"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / nl if nl else None
                    r_["traffic_source"] = ("REPLAYED from the committed profile " + pm["_file"] + " (rocprofv3 --pmc FETCH_SIZE / "
                                            "WRITE_SIZE passes of this command on the serialized step; FETCH_SIZE doubled per the "
                                            "gfx950 correction; not observed in this run): HBM bytes per launch")
            ut = load_profile_json(f"r05_{prec}_mfma_util.json") or load_profile_json(f"r04_{prec}_mfma_util.json")
            if ut:
                for r_, key in ((roof, "conv3x3"), (roof_w, "wgrad")):
                    r_["mfma_busy_frac"] = (ut.get(key + "_group") or ut.get(key, {})).get("mfma_busy_frac")
                    r_["util_source"] = "REPLAYED from the committed profile " + ut["_file"] + " (not observed in this run): " + str(ut.get("note"))
            # (no replayed clocks: the shader clock under a dense MFMA stream is OBSERVED in this run, before and after
            # the serialized pass -- roofline.mfma_calibration)
        if sf == 64 and args.precision in ("bf16", "h3p", "h3f"):
            replay(rl, main["roofline_wgrad"], args.precision)
        if parity is not None and args.parity_precision in ("bf16", "h3p", "h3f"):
            replay(parity["roofline"], parity["roofline_wgrad"], args.parity_precision)
        workload = ("BASELINE configs[1]: U-Net (depth 5, 64 filters) train step, batch 32 x 4x256x256 per GPU"
                    if sf == 64 else
                    f"BASELINE configs[4]: wide U-Net (depth 5, {sf} filters) train step, batch {args.batch} x 4x256x256 "
                    "per GPU" + (", on-GPU add_noise/flip augment" if args.gpu_augment else ""))
        if world > 1:
            be = dist.get_backend()
            workload += (f", data-parallel over {world} ranks ({'RCCL over xGMI' if be == 'nccl' else be + ' (rehearsal backend)'} "
                         "gradient exchange overlapped with backward)")
        out = {
            "metric": "echogram patches/sec (4ch 256x256), training step (fwd+weighted CE+bwd+SGD)",
            "value": main["train_patches_per_s"], "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": main["dtype"], "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * args.batch, "patch": [4, 256, 256],
                       "precision": args.precision, "start_filts": sf, "parallelism": f"dp{world}",
                       "ranks": world, "backend": (dist.get_backend() if world > 1 else None), "rank_census": census,
                       "launcher": "self-spawned" if os.environ.get("CRIMAC_SELF_LAUNCHED") == "1" else
                                   ("torchrun" if world > 1 else "single process")},
            "per_gpu_patches_per_s": main["train_patches_per_s"] / world,
            "train_tflops": main["train_tflops"],
            "infer_patches_per_s": main.get("infer_patches_per_s"),
            "infer_tflops": main.get("infer_tflops"),
            "final_loss": main["final_loss"], "loss_scale": main["loss_scale"], "skipped_steps": main["skipped_steps"],
            "golden_parity": main["golden_parity"],
            "roofline": rl, "roofline_wgrad": main["roofline_wgrad"],
        }
        if main.get("train_loop") is not None:
            out["train_loop"] = main["train_loop"]
        if main.get("train_loop_raw") is not None:
            out["train_loop_raw"] = main["train_loop_raw"]
        if main.get("exchange") is not None:
            out["exchange"] = main["exchange"]
        if parity is not None:
            out["parity_mode"] = parity
        if tiled is not None:
            out["tiled"] = tiled
        if wide is not None:
            out["wide_fp16"] = wide
        if world == 1 and not args.no_cpu_baseline:
            def parity32(x32, ref_logits):
                """The parity precision on the 32 crops the oracle has just been timed on: logits rel, argmax flips and the
                oracle's own top-2 margin at every flipped pixel (a flip below ~2e-6 is the oracle's fp32 round-off)."""
                import torch
                import crimac_classifiers_unet_amd as pkg
                from crimac_classifiers_unet_amd import synth
                m = pkg.UNet_Baseline(3, 4, precision=args.parity_precision)
                m.load_state_dict(synth.synth_state_dict(seed=0))
                m.to(dev).eval()
                with torch.no_grad():
                    got = m(x32.to(dev)).float().cpu()
                diff = got.argmax(1) != ref_logits.argmax(1)
                top2 = ref_logits.topk(2, dim=1).values
                margins = (top2[:, 0] - top2[:, 1])[diff]
                # the criterion of tests/test_gpu_h3p.py (test_network_batch32_full_size_matches_oracle, which runs these
                # very crops as its seed-1 case): every flipped pixel sits where the oracle's own two top logits are
                # closer than its fp32 round-off, and such ties are rare
                tie, frac = 2e-6, 1e-5
                ok = bool((margins < tie).all()) and int(diff.sum()) <= frac * diff.numel()
                return {"precision": args.parity_precision, "crops": "32 x 4 x 256 x 256 (synthetic, seed 1) vs the CPU oracle",
                        "pixels": int(diff.numel()), "logits_rel": float((got - ref_logits).abs().max() / ref_logits.abs().max()),
                        "argmax_flips": int(diff.sum()), "oracle_top2_margin_at_flips": [float(v) for v in margins[:8]],
                        "max_oracle_margin_at_flips": float(margins.max()) if margins.numel() else 0.0,
                        "criterion": f"every flip at an oracle top-2 margin < {tie:g} and flips <= {frac:g} of the pixels "
                                     "(the assertion of tests/test_gpu_h3p.py on the same crops)",
                        "identical_masks_up_to_oracle_ties": ok}
            cb = cpu_baseline(parity32 if (parity is not None and args.start_filts == 64) else None)
            p32 = cb.pop("parity_batch32")
            if p32 is not None:
                out["parity_mode"]["batch32_vs_oracle"] = p32
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    from crimac_classifiers_unet_amd import launch
    if args.gpus > 1 and not launch.launched_by_torchrun():
        # parent: no GPU call in this process -- spawn one rank per GPU and relay rank 0's line
        print(f"[bench] no torchrun environment: spawning {args.gpus} ranks (one per GPU)", file=sys.stderr, flush=True)
        rc, out0 = launch.spawn_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], args.gpus)
        # rank 0's stdout: the result line (a backend may chat on stdout too -- gloo does -- that goes to stderr)
        for line in out0.splitlines():
            (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
        sys.stdout.flush()
        sys.exit(rc)
    run_rank(args)


if __name__ == "__main__":
    main()
