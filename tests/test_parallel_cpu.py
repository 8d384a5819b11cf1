"""CPU, world size 2, gloo: the N>1 host path (gradient exchange, inference shard/gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from crimac_classifiers_unet_amd import parallel
    w, r, _ = parallel.init_distributed(backend="gloo")
    assert (w, r) == (world, rank)
    # gradient exchange over a flat buffer with buckets that do not divide it evenly
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    scale = parallel.GradSync(bucket_mb=0.001)(flat)          # 262-element buckets
    expect = torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1))
    ok_grad = bool(torch.equal(flat, expect)) and abs(scale - 1.0 / world) < 1e-12
    # overlapped form: ranges launched out of order while "backward" is still running, rest at the end
    gs = parallel.GradSync(bucket_mb=0.001)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    gs.launch(flat, 700, 1000)
    gs.launch(flat, 300, 700)
    assert gs.pending_ranges(1000) == [(0, 300)]
    try:
        gs.launch(flat, 650, 800)
        ok_grad = False
    except RuntimeError:
        pass
    scale = gs(flat)
    ok_grad = ok_grad and bool(torch.equal(flat, expect)) and abs(scale - 1.0 / world) < 1e-12 \
        and gs.pending_ranges(1000) == [(0, 1000)]
    # reduce-scatter + all-gather spelling of the same exchange (even buckets take it, the ragged tail all-reduces)
    try:
        gs = parallel.GradSync(bucket_mb=0.001, algo="rs_ag")
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        scale = gs(flat)
        ok_grad = ok_grad and bool(torch.equal(flat, expect)) and abs(scale - 1.0 / world) < 1e-12
    except RuntimeError as e:            # a backend without reduce_scatter_tensor must say so, not corrupt data
        ok_grad = ok_grad and ("reduce_scatter" in str(e).lower() or "not supported" in str(e).lower()
                               or "unsupported" in str(e).lower())
    # per-range completion (the optimiser step of a range runs while later ranges are still being exchanged)
    for algo in ("all_reduce", "rs_ag"):
        gs = parallel.GradSync(bucket_mb=0.001, algo=algo)
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        gs.launch(flat, 600, 1000)
        gs.launch(flat, 0, 600)
        sc = gs.finish_range(600, 1000)
        ok_grad = ok_grad and bool(torch.equal(flat[600:], expect[600:])) and abs(sc - 1.0 / world) < 1e-12
        gs.finish_range(0, 600)
        ok_grad = ok_grad and bool(torch.equal(flat, expect))
        try:
            gs.finish_range(1, 2)
            ok_grad = False
        except RuntimeError:
            pass
        gs.finish()
        ok_grad = ok_grad and gs.pending_ranges(1000) == [(0, 1000)]
    # SGD with grad_scale=1/world on the summed gradient == SGD on the mean gradient
    # inference: 7 "patches", each rank computes its shard, all-gather restores patch order
    n = 7
    idx = parallel.shard_indices(n, rank, world)
    local = torch.tensor([[10.0 * i, 10.0 * i + 1] for i in idx]).reshape(-1, 2)        # (a rank may own no patch)
    full = parallel.gather_shards(local, n)
    ok_gather = bool(torch.equal(full, torch.tensor([[10.0 * i, 10.0 * i + 1] for i in range(n)])))
    sums = parallel.all_reduce_scalars(torch.tensor([1.0 + rank, 2.0]))
    ok_sums = sums.tolist() == [sum(1.0 + r for r in range(world)), 2.0 * world]
    results[rank] = (ok_grad, ok_gather, ok_sums)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_world_size_n_gloo(world):
    """world 8 = the node the scaling bench runs on (one rank per GPU): rendezvous, bucketed exchange in both spellings,
    out-of-order ranges, per-range completion, patch gather with ranks that own nothing."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        assert dict(results) == {r: (True, True, True) for r in range(world)}


def test_single_process_is_a_noop():
    from crimac_classifiers_unet_amd import parallel
    flat = torch.ones(10)
    assert parallel.GradSync()(flat) == 1.0 and bool(torch.equal(flat, torch.ones(10)))
    t = torch.arange(6.0).reshape(3, 2)
    assert parallel.gather_shards(t, 3) is t
