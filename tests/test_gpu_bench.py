"""bench.py end to end on a GPU: the JSON contract of the single-GPU line, and a two-rank rehearsal of
the multi-GPU path (both ranks on GPU 0, gloo in place of RCCL; see RRT_BENCH_REHEARSAL in bench.py)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def last_json(text):
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp-side", "8"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "Msamples/s"
    assert d["config"]["samples_per_step"] == 500 * 500 * 64 and d["vs_baseline"] is None and d["dtype"] == "f32"
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"} <= set(rf) and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and d["value"] > 20 * cb["value"]


def test_two_rank_rehearsal():
    """`python bench.py --gpus 2` with no launcher in front: the script starts its two ranks itself (fresh children,
    before any GPU call).  Strong scaling: ONE 64-spp image, 32 sample indices per rank."""
    env = dict(os.environ, RRT_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--spp-side", "8", "--workload", "smallpt"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "strong" and "rehearsal" in d
    assert d["config"]["samples_per_step"] == 1024 * 768 * 64  # the two ranks' 32 sample indices of every pixel
    assert "cpu_baseline" not in d  # timed on rank 0 at N = 1 only


def test_two_rank_rehearsal_weak_under_a_launcher():
    """The way the driver starts N > 1: torch.distributed.run in front.  Weak scaling: 64 sample indices per rank."""
    env = dict(os.environ, RRT_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--spp-side", "8", "--workload", "smallpt", "--scaling", "weak"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "rehearsal" in d
    assert d["config"]["samples_per_step"] == 2 * 1024 * 768 * 64


RCCL_PROBE = r'''
import os, sys
sys.path.insert(0, os.environ["RRT_ROOT"])
import numpy as np, torch, torch.distributed as dist
from rustraytracer_amd import api, config
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))  # what bench.py does for N > 1
be = api.Backend(0)
be.set_stream(torch.cuda.current_stream().cuda_stream)
sc = api.Scene(config.load_scene(os.path.join(os.environ["RRT_ROOT"], "configs", "smallpt", "scene.toml")))
rc = config.stratified_config(64, 48, 4, 4)
acc = torch.zeros((48, 64, 3), dtype=torch.float32, device="cuda")
be.render_accumulate(sc, rc.render_desc(), acc)
before = acc.clone()
dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)   # RCCL kernels on torch's stream, after the render on the same stream
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(acc, before) and float(t.item()) == 1.5 and float(acc.sum()) > 0
dist.destroy_process_group()
print("rccl probe ok")
'''


def test_rccl_process_group_of_one():
    """The multi-GPU bench path minus the second GPU: an RCCL ("nccl") process group of one rank on this box, the
    sum-reduce and the max all-reduce bench.py issues, on an accumulator the library has just written on torch's stream."""
    env = dict(os.environ, RRT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", RCCL_PROBE], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0 and "rccl probe ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
