"""N > 1 path on CPU: sample-index sharding + sum-reduce of the accumulators, world_size 2, gloo."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT
from rustraytracer_amd import distributed as D


def test_shard_ranges_tile_the_sample_range():
    for total in (1, 7, 64, 16384, 16385):
        for world in (1, 2, 3, 4, 8):
            edges = [D.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_weak_scaling_strata():
    assert D.weak_scaling_strata(128, 128, 1) == (128, 128)
    assert D.weak_scaling_strata(128, 128, 2) == (256, 128)
    assert D.weak_scaling_strata(128, 128, 4) == (256, 256)
    assert D.weak_scaling_strata(128, 128, 8) == (512, 256)
    x, y = D.weak_scaling_strata(32, 32, 6)
    assert x * y == 6 * 1024


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from rustraytracer_amd import config, distributed as D
    from oracle import oracle_py as O   # CPU stand-in for the per-rank renderer in this test
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    flat = config.load_scene(os.path.join(%r, "configs", "cornell_box", "scene.toml"))
    rc = config.stratified_config(16, 16, 4, 4, seed=rank + 1)
    lo, hi = D.shard_range(rc.spp, rank, world)
    # each rank contributes (its share of the samples) x (a render of the scene): what matters here is
    # that the reduce adds the two accumulators element-wise on rank 0 and leaves the shard plan intact
    rgb, st = O.render_accumulate(flat, rc.render_desc())
    mine = torch.from_numpy((rgb * (hi - lo) / rc.spp).astype(np.float32))
    keep = mine.clone()
    D.reduce_sum(mine, dst=0)
    gathered = [torch.zeros_like(keep) for _ in range(world)]
    dist.all_gather(gathered, keep)
    if rank == 0:
        expect = sum(gathered)
        assert torch.allclose(mine, expect, rtol=1e-6), "reduce != sum of shards"
        assert (hi - lo) * world == rc.spp
        print("GLOO_OK", float(mine.mean()))
    dist.destroy_process_group()
""")


def test_two_rank_reduce_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO_OK" in out.stdout
