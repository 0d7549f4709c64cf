"""CPU: ``python bench.py --gpus N`` without a launcher starts its own N ranks (it used to exit asking for
torch.distributed.run, so the first contact of the scaling run with hardware would have been rc != 0).  The launch path alone
is exercised here - INERF_BENCH_LAUNCH_PROBE=1 makes every rank stop after the rendezvous, before anything touches a GPU."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_command_is_the_drivers_torchrun_line():
    bench = _bench_module()
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "3"], 29511)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                   "--master-port", "29511", BENCH, "--gpus", "8", "--steps", "20", "--warmup", "3"]


def _run(args, **env):
    e = dict(os.environ, INERF_BENCH_LAUNCH_PROBE="1", **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=600, cwd=REPO)


def test_plain_invocation_with_gpus_2_starts_two_ranks_and_prints_one_line():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "1"])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout          # ONE line on stdout: what gloo and the launcher say goes to stderr
    rec = json.loads(lines[0])
    assert rec == {"launch_probe": True, "n_gpus": 2, "ranks_seen": 2, "steps": 1, "warmup": 1, "local_rank": 0}


def test_a_failing_rank_fails_the_plain_invocation():
    out = _run(["--gpus", "2"], INERF_BENCH_LAUNCH_PROBE_RC="3")
    assert out.returncode != 0


def test_single_gpu_invocation_does_not_spawn():
    out = _run(["--gpus", "1", "--steps", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1
