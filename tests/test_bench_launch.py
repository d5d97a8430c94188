"""bench.py's own launcher on a CPU box: `python bench.py --gpus N` without torchrun in front must start N fresh
ranks itself, before any GPU call, and relay rank 0's line.  RRT_BENCH_REHEARSAL=plumbing takes the renderer out
(no GPU here): what runs is the launch, the gloo process group, the shard plan, the reduce and the timing protocol."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run_bench(*argv, env_extra=None):
    env = dict(os.environ, RRT_BENCH_REHEARSAL="plumbing")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300,
                       cwd=ROOT, env=env)
    return r


def test_self_launch_two_ranks_strong():
    r = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--spp-side", "8")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly one line reaches the caller's stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    # one image, one spp: 64 samples per pixel in total, 32 sample indices per rank
    assert d["config"]["samples_per_step"] == 500 * 500 * 64
    assert "[0, 32)" in d["config"]["workload"] and "64 spp in total" in d["config"]["workload"]
    assert "rehearsal" in d and "cpu_baseline" not in d
    # the line explains itself: every rank's kernel time and reduce time per step, and the path the library reports
    assert len(d["per_rank_kernel_ms"]) == 2 and len(d["per_rank_reduce_ms"]) == 2
    assert d["reduce_ms"] == max(d["per_rank_reduce_ms"]) and d["reduce_ms"] > 0
    assert d["config"]["accel_used"] == "linear" and d["config"]["pipeline_used"] == "fused"


def test_self_launch_weak_scaling_three_ranks():
    r = run_bench("--gpus", "3", "--steps", "1", "--warmup", "0", "--spp-side", "4", "--scaling", "weak", "--workload", "smallpt")
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3 and d["scaling"] == "weak"
    assert d["config"]["samples_per_step"] == 3 * 1024 * 768 * 16  # 16 sample indices per rank, 48 per pixel


def test_failed_rank_gives_nonzero_exit():
    # more ranks than sample indices: rank 1 has nothing to trace and refuses; the launcher must report failure
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--spp-side", "1")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_without_gpu_the_bench_refuses_to_run():
    # no rehearsal switch, no GPU in this container: the product path has no CPU fallback
    import torch

    if torch.cuda.is_available():
        return
    r = run_bench("--steps", "1", "--warmup", "0", "--spp-side", "2", env_extra={"RRT_BENCH_REHEARSAL": ""})
    assert r.returncode != 0 and "needs a GPU" in r.stderr
