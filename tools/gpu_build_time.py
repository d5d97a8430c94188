"""The three tree builders on the million-triangle scene (configs/mesh_1m): time of the first call that needs the tree
(rrt_stats.tree_build_ms: host preparation, upload, build, leaf records), depth, and traversal steps per ray and rate
of a reduced frame through the wavefront pipeline.  usage: gpu_build_time.py [W H S]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_c5_assets
make_c5_assets.main()
from rustraytracer_amd import api, config
W, H, S = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 8)
be = api.Backend(0)
flat = config.load_scene(os.path.join(ROOT, "configs/mesh_1m/scene.toml"))
print("primitives", flat.n_prims, flush=True)
small = config.stratified_config(64, 64, 2, 2)
rc = config.stratified_config(W, H, S, S)
for name, accel in (("sah_gpu", api.ACCEL_SAH_GPU), ("sah (host)", api.ACCEL_SAH), ("lbvh", api.ACCEL_LBVH), ("sah_gpu again", api.ACCEL_SAH_GPU)):
    sc = api.Scene(flat)
    t0 = time.time()
    be.render_accumulate(sc, small.render_desc(accel=api.ACCEL_LINEAR, pipeline=api.PIPELINE_FUSED, sample_end=1) if False else small.render_desc(accel=accel))
    first = time.time() - t0
    _, st0 = be.render_accumulate(api.Scene(flat), small.render_desc(accel=accel))
    best = None
    for rep in range(2):
        _, st = be.render_accumulate(sc, rc.render_desc(accel=accel))
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    print("%-14s first call %.3f s, tree build %.1f ms (second scene %.1f ms), depth %d, node/ray %.2f prim/ray %.2f, %.0f Mrays/s" % (
        name, first, st0["tree_build_ms"] if False else 0.0, st0["tree_build_ms"], best["tree_depth"], best["node_tests"] / best["rays"],
        best["prim_tests"] / best["rays"], best["rays"] / best["kernel_ms"] / 1e3), flush=True)
