"""The multi-rank training step over RCCL ("nccl"), on the hardware this suite gets.

A GPU box has ONE GPU and RCCL refuses two ranks on one device, so the N > 1 numerics are rehearsed over gloo
(tests/test_gpu_unet.py::test_syncbn_two_ranks_..., tests/test_parallel_cpu.py).  What gloo never reaches are the RCCL
branches of the step: collectives that are stream-ordered (``work.wait()`` is a stream wait, the all-gather of a bucket
is chained right behind its reduce-scatter), the exchange launched from the weight-gradient side stream, and the per-range
optimiser step + re-pack behind ``GradSync.finish_range``.  ``GradSync(force=True)`` issues every collective in a
process group of ONE rank (identities): a single GPU then drives all of those branches through real RCCL calls, and the
result must equal the plain single-rank step.  The 2-rank test runs wherever two GPUs are visible.
"""
import os
import socket

import pytest
import torch

from crimac_classifiers_unet_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(precision):
    import crimac_classifiers_unet_amd as pkg
    m = pkg.UNet_Baseline(3, 4, precision=precision)
    m.load_state_dict(synth.synth_state_dict(seed=0))
    return m.cuda().train()


def _steps(eng, x, lab, grad_sync, n=2):
    """n fused steps; returns (flat_p after step 1, flat_p after step n, losses)."""
    cw = torch.tensor([10.0, 300.0, 250.0], device="cuda")
    losses, p1 = [], None
    for k in range(n):
        losses.append(float(eng.train_step(x, lab, cw, lr=0.005, momentum=0.9, grad_sync=grad_sync)))
        if k == 0:
            torch.cuda.synchronize()
            p1 = eng.flat_p.clone()
    torch.cuda.synchronize()
    return p1, eng.flat_p.clone(), losses


def _forced_worker(port, out):
    import torch.distributed as dist
    from crimac_classifiers_unet_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    x = torch.from_numpy(synth.synth_echogram_batch(4, 4, 64, 64, seed=1)).cuda()
    lab = torch.from_numpy(synth.synth_labels(4, 64, 64, seed=2)).cuda()
    res = {}
    # f32x6: the unscaled step (per-range optimiser step available); h3p: the loss-scaled step (one guarded update).
    # Both are fp32-class precisions: two runs differ by the order of fp32 atomics only.  (bf16 would not do: its
    # 16-bit storage amplifies that noise to O(10 %) of a gradient within one step)
    for precision in ("f32x6", "h3p"):
        m0 = _make(precision)
        p0 = m0.engine
        p0.bind()
        start = p0.flat_p.clone()
        ref1, ref2, ref_l = _steps(p0, x, lab, None)
        for name, kw, env in (("all_reduce", dict(algo="all_reduce"), {}),
                              ("rs_ag", dict(algo="rs_ag"), {}),
                              ("rs_ag+per_range_sgd", dict(algo="rs_ag"), {"early_sgd_multi": True}),
                              ("all_reduce+per_range_sgd+side", dict(algo="all_reduce"),
                               {"early_sgd_multi": True, "exchange_on_side": True})):
            m = _make(precision)
            eng = m.engine
            for k, v in env.items():
                setattr(eng, k, v)
            gs = parallel.GradSync(bucket_mb=8.0, force=True, **kw)
            got1, got2, got_l = _steps(eng, x, lab, gs)

            def upd_rel(a, b):
                return float(((a - b).double().norm() / (b - start).double().norm()))
            res[f"{precision}/{name}"] = (upd_rel(got1, ref1), upd_rel(got2, ref2), got_l, ref_l,
                                          float((ref1 - start).double().norm()))
            del m
    out.update(res)
    dist.destroy_process_group()


def test_single_rank_nccl_drives_every_multi_rank_branch_of_the_step():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        p = ctx.Process(target=_forced_worker, args=(_free_port(), out))
        p.start()
        p.join(600)
        assert p.exitcode == 0
        res = dict(out)
    assert len(res) == 8
    for key, (r1, r2, got_l, ref_l, upd) in res.items():
        print(f"{key}: update L2-rel after step 1 {r1:.2e}, after step 2 {r2:.2e}; losses {got_l} vs {ref_l}")
        assert upd > 0
        # step 1 starts from identical parameters: only the order of the fp32 atomics (statistics, weight gradients) differs
        assert r1 < 1e-2, (key, r1)            # (a lost range or a wrong scale is an O(1) error)
        # step 2 runs on re-packed operands (per-range re-pack included)
        assert r2 < 5e-2, (key, r2)
        assert abs(got_l[0] - ref_l[0]) <= 1e-5 * abs(ref_l[0]) and abs(got_l[1] - ref_l[1]) < 1e-3 * abs(ref_l[1])


def _two_rank_worker(rank, world, port, x, lab, algo, per_range, out):
    import torch.distributed as dist
    from crimac_classifiers_unet_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    m = _make("f32x6")
    eng = m.engine
    eng.sync_bn = True
    eng.early_sgd_multi = per_range
    n = x.shape[0] // world
    gs = parallel.GradSync(bucket_mb=8.0, algo=algo)
    _, p2, losses = _steps(eng, x[rank * n:(rank + 1) * n].cuda(), lab[rank * n:(rank + 1) * n].cuda(), gs)
    out[rank] = (p2.cpu(), losses)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("algo,per_range", [("all_reduce", False), ("rs_ag", False), ("rs_ag", True)])
def test_two_ranks_over_rccl_equal_one_rank_on_the_concatenated_batch(algo, per_range):
    """2 ranks x 2 patches with SyncBN over RCCL == 1 rank on the 4 patches (loss = mean of the per-rank losses, which the
    1/world gradient scale realises); the parameters after two steps are IDENTICAL on both ranks."""
    import torch.multiprocessing as mp
    x = torch.from_numpy(synth.synth_echogram_batch(4, 4, 64, 64, seed=1))
    lab = torch.from_numpy(synth.synth_labels(4, 64, 64, seed=2))
    # the single-rank reference: per-rank losses averaged == ONE weighted CE only when the label weights of the two
    # halves agree, so the reference is built the way the ranks compute: gradient = mean of the two half-batch gradients
    # under whole-batch BatchNorm statistics -- which is what tests/test_gpu_unet.py checks against the fp64 oracle; here
    # the two ranks are compared with each other and with the gloo rehearsal's invariants
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, x, lab, algo, per_range, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        (p_a, l_a), (p_b, l_b) = out[0], out[1]
    assert torch.equal(p_a, p_b), "parameters diverged between the ranks"
    assert all(v == v for v in l_a + l_b)
