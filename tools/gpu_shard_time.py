"""Kernel time of one rank's share of the headline frame when N GPUs split it (sample indices [0, 16384 / N))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustraytracer_amd import api, config, distributed as D
be = api.Backend(0)
flat = config.load_scene(os.path.join(ROOT, "configs/cornell_box/scene.toml"))
sc = api.Scene(flat)
rc = config.stratified_config(500, 500, 128, 128)
full = None
for n in (1, 2, 4, 8, 16):
    lo, hi = D.shard_range(rc.spp, 0, n)
    best = 1e9
    for rep in range(6):
        _, st = be.render_accumulate(sc, rc.render_desc(sample_begin=lo, sample_end=hi))
        best = min(best, st["kernel_ms"])
    full = full or best
    print("N=%2d: %5d spp per GPU, kernel %.3f ms, %.0f Msamples/s per GPU, efficiency of the kernel alone %.3f" % (
        n, hi - lo, best, st["samples"] / best / 1e3, full / n / best), flush=True)
