"""`python bench.py --gpus N` launched PLAINLY (no torch.distributed.run around it, no RANK in the
environment) must become its own launcher and bring N ranks up to the communicator's set-up -- the
first real 8-GPU run must not end in a SystemExit.  CPU-only: WM_BENCH_DRY_LAUNCH=1 stops every rank
right after the ranks have found each other (gloo all-reduce of the rank numbers)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)


def test_plain_launch_with_two_gpus_reaches_the_rendezvous():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WM_BENCH_DRY_LAUNCH": "1"})
    assert r.returncode == 0, r.stdout[-2000:]
    assert "needs torch.distributed.run" not in r.stdout
    for rank in (0, 1):
        assert "dry-launch rank %d of 2: ranks sum 1" % rank in r.stdout, r.stdout[-2000:]


def test_plain_launch_with_one_gpu_stays_in_process():
    # N = 1 never relaunches: one rank, no launcher
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WM_BENCH_DRY_LAUNCH": "1"})
    assert r.returncode == 0, r.stdout[-2000:]
    assert "dry-launch rank 0 of 1: ranks sum 0" in r.stdout, r.stdout[-2000:]
