"""bench.py --gpus N starts its own ranks: exercised here on the CPU with --dry (gloo, no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "3", "--warmup", "1", *extra],
                       env=env, capture_output=True, text=True, timeout=300)
    return p


def test_self_launch_two_ranks():
    p = _run("--gpus", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["dry"] is True
    assert line["steps"] == 3 and line["warmup"] == 1
    assert line["max_elapsed_s"] >= 0.02       # the MAX over ranks: rank 1 sleeps 20 ms


def test_single_rank_needs_no_launcher():
    p = _run("--gpus", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["world_size"] == 1


def test_world_size_mismatch_is_an_error():
    p = _run("--gpus", "2", env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_full_step_harness_two_ranks_dry():
    """bench.py --gpus 2 --dry --workload full_step: the config-5 harness through self_launch over gloo -- the line proves
    the process group saw both ranks (world_size, a SUM of ones), carries comm_ms, and the replicas' weights agree."""
    p = _run("--gpus", "2", "--workload", "full_step")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["dry"] is True and line["n_gpus"] == 2 and line["world_size"] == 2 and line["ranks_summed"] == 2
    assert line["backend"] == "gloo" and line["config"]["workload"] == "full_step"
    assert line["comm_ms"] > 0 and line["buckets"] >= 3 and line["grad_bytes_all_reduced"] == 4 * (40_000 + 3_001 + 777)
    assert line["replica_weight_spread"] == 0.0


import pytest


@pytest.mark.gpu
def test_two_ranks_with_real_kernels_on_one_gpu():
    """The N > 1 path of bench.py with the HIP kernels running: two self-launched ranks share cuda:0 over gloo (RCCL does not
    take two ranks on one device).  One JSON line from rank 0, value = both ranks' pixels over the MAX of their times, the
    process group's own record of its size."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "20",
                        "--warmup", "3", "--no-cpu", "--cold-steps", "0", "--pipelined", "0", "--spin-up", "0.05"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 20
    pg = line["process_group"]
    assert pg["world_size"] == 2 and pg["ranks_summed"] == 2 and pg["backend"] == "gloo" and pg["comm_ms"] > 0
    px = 2 * 12 * 2 * 192 * 640 * 20                       # both ranks' warped pixels in the timed steps
    assert abs(line["value"] - px / (line["ms_per_step"] * 1e-3 * 20) / 1e9) < 1e-3 * line["value"]
    assert "rehearsal" in line and "also" not in line and "cpu_baseline" not in line
