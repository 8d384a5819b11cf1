"""Self-launch of one process per GPU on one node (the reference has no distributed code).

``python bench.py --gpus N`` must measure N GPUs even when it is not started by ``torchrun``: the parent
process spawns N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their
environment -- exactly what ``python -m torch.distributed.run --nproc-per-node N`` would set), forwards their
stderr, and returns rank 0's stdout.  The parent never touches the GPU (no HIP call, no
``torch.cuda.is_available()``): a process that has initialised the GPU must not start or replace programs on
this pool, and the children must find every device unclaimed.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time


def launched_by_torchrun() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(argv, n_ranks, extra_env=None, timeout=None):
    """Run ``argv`` (a full command line) as ``n_ranks`` processes, one per local rank.

    Returns ``(returncode, rank0_stdout)``; ``returncode`` is the first non-zero exit code of any rank
    (the other ranks are then terminated by PID).  Ranks > 0 have their stdout discarded (rank 0 prints
    the result line); every rank's stderr goes to this process's stderr.
    """
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks),
                    "LOCAL_WORLD_SIZE": str(n_ranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "CRIMAC_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this pool)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    t0 = time.time()
    rc = 0
    out0 = ""
    kill_at = None               # after a SIGTERM the survivors get GRACE_S seconds, then SIGKILL (a rank stuck in the
    GRACE_S = 10.0               # driver or in a collective may ignore SIGTERM: the parent must not wait for it for ever)
    try:
        # rank 0's stdout is small (one JSON line): read it to the end, then reap everyone
        alive = list(range(n_ranks))
        import threading
        buf = []
        reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        while alive:
            for r in list(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.remove(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[launch] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    for q in alive:
                        procs[q].send_signal(signal.SIGTERM)
                    kill_at = kill_at or time.time() + GRACE_S
            if timeout is not None and time.time() - t0 > timeout:
                rc = rc or 124
                print(f"[launch] timeout after {timeout} s; stopping all ranks", file=sys.stderr)
                for q in alive:
                    procs[q].send_signal(signal.SIGTERM)
                timeout = None
                kill_at = kill_at or time.time() + GRACE_S
            if kill_at is not None and alive and time.time() > kill_at:
                print(f"[launch] ranks {alive} ignored SIGTERM for {GRACE_S:.0f} s: killing them", file=sys.stderr)
                for q in alive:
                    procs[q].kill()                    # (by PID: children of this process only)
                for q in alive:
                    try:
                        procs[q].wait(timeout=5)
                    except subprocess.TimeoutExpired:
                        pass
                rc = rc or 137
                break
            time.sleep(0.05)
        reader.join(timeout=5)
        out0 = buf[0] if buf else ""
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc, out0
