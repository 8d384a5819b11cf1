"""bench.py --gpus N must start N ranks by itself when it is not launched by torchrun.

CPU rehearsal: `--spawn-selftest` runs only the launcher + rendezvous + one all-reduce (gloo here, RCCL on a GPU
node) and prints the census; the parent process never touches the GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CRIMAC_SELF_LAUNCHED")}
    env["CRIMAC_DIST_BACKEND"] = "gloo"
    return env


def test_bench_gpus2_spawns_two_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-selftest"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["ranks_counted"] == 2 and out["self_launched"] is True
    assert out["backend"] == "gloo"


def test_bench_under_torchrun_env_does_not_respawn():
    env = _clean_env()
    env.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn-selftest"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["self_launched"] is False


def test_failed_rank_stops_the_job():
    from crimac_classifiers_unet_amd import launch
    code = "import os,sys,time\nif os.environ['RANK']=='1': sys.exit(7)\ntime.sleep(60)\n"
    rc, _ = launch.spawn_ranks([sys.executable, "-c", code], 2, timeout=30)
    assert rc == 7


def test_bench_gpus8_spawns_eight_ranks():
    """The scaling bench's largest case, rehearsed over gloo on the CPU: 8 self-spawned ranks, one free port, census 8,
    clean exit of all of them."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--spawn-selftest"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 8 and out["ranks_counted"] == 8 and out["self_launched"] is True
    assert out["backend"] == "gloo" and out["gradsync_ok"] is True


def test_bench_under_torchrun_with_eight_ranks():
    """The driver's own launch line for N = 8 (python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8):
    bench.py is then one of the ranks and must not spawn again."""
    from crimac_classifiers_unet_amd import launch
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(launch.free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8",
           "--spawn-selftest"]
    p = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["ranks_counted"] == 8 and out["self_launched"] is False and out["gradsync_ok"] is True


def test_failed_rank_stops_the_job_at_eight_ranks():
    from crimac_classifiers_unet_amd import launch
    code = "import os,sys,time\nif os.environ['RANK']=='5': sys.exit(9)\ntime.sleep(60)\n"
    rc, _ = launch.spawn_ranks([sys.executable, "-c", code], 8, timeout=30)
    assert rc == 9
