"""Wavefront pipeline on the triangle-mesh test scene of N triangles (tests/test_gpu_lbvh.py), both trees:
   gpu_mesh_rate.py N [reps]    -> Mrays/s and steps per ray of every repetition"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rustraytracer_amd import api, config
import test_gpu_lbvh as T
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
be = api.Backend(0)
sc = api.Scene(T.mesh_scene(n))
rc = config.stratified_config(1024, 1024, 16, 16)
for name, accel in (("cluster", api.ACCEL_LBVH), ("area", api.ACCEL_SAH)):
    out = []
    for rep in range(reps):
        rgb, st = be.render_accumulate(sc, rc.render_desc(accel=accel, pipeline=api.PIPELINE_WAVEFRONT))
        out.append("%.0f" % (st["rays"] / st["kernel_ms"] / 1e3))
    print("%d triangles, %s tree: %s Mrays/s, node/ray %.2f prim/ray %.2f" % (n, name, " ".join(out), st["node_tests"] / st["rays"], st["prim_tests"] / st["rays"]), flush=True)
