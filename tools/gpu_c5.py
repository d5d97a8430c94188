"""Throughput of the LBVH path on the million-triangle scene (configs/mesh_1m), reduced image size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_c5_assets
make_c5_assets.main()
from rustraytracer_amd import api, config
W, H, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
be = api.Backend(0)
t0 = time.time(); flat = config.load_scene(os.path.join(ROOT, "configs/mesh_1m/scene.toml")); print("load %.2fs prims %d" % (time.time() - t0, flat.n_prims), flush=True)
t0 = time.time(); sc = api.Scene(flat); print("scene create %.2fs" % (time.time() - t0), flush=True)
rc = config.stratified_config(W, H, S, S)
rc.max_depth = int(os.environ.get("C5_MAX_DEPTH", "6"))
ACCEL = {"lbvh": api.ACCEL_LBVH, "sah": api.ACCEL_SAH, "sah_gpu": api.ACCEL_SAH_GPU, "auto": api.ACCEL_AUTO}[os.environ.get("RRT_C5_ACCEL", "auto")]  # which tree
t0 = time.time(); rgb, st = be.render_accumulate(sc, config.stratified_config(64, 64, 2, 2).render_desc(accel=ACCEL)); print("first call (build) %.2fs" % (time.time() - t0), flush=True)
for rep in range(2):
    rgb, st = be.render_accumulate(sc, rc.render_desc(accel=ACCEL))
    print("render:", {k: st[k] for k in ("samples", "rays", "prim_tests", "node_tests", "kernel_ms")},
          "Msamples/s %.1f Mrays/s %.1f node/ray %.1f prim/ray %.2f" % (st["samples"] / st["kernel_ms"] / 1e3, st["rays"] / st["kernel_ms"] / 1e3,
          st["node_tests"] / st["rays"], st["prim_tests"] / st["rays"]), flush=True)
img = be.postprocess(rgb, rc.spp, rc.post_desc())
from PIL import Image
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
Image.fromarray(img).save(os.path.join(ROOT, "gpurun_out", "c5.png"))
