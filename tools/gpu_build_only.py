"""One build of a tree of the million-triangle scene and nothing else (for rocprofv3 --kernel-trace --stats):
gpu_build_only.py [sah_gpu|sah|lbvh]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_c5_assets
make_c5_assets.main()
from rustraytracer_amd import api, config
accel = {"sah_gpu": api.ACCEL_SAH_GPU, "sah": api.ACCEL_SAH, "lbvh": api.ACCEL_LBVH}[sys.argv[1] if len(sys.argv) > 1 else "sah_gpu"]
be = api.Backend(0)
flat = config.load_scene(os.path.join(ROOT, "configs/mesh_1m/scene.toml"))
small = config.stratified_config(16, 16, 1, 1)
for rep in range(2):
    _, st = be.render_accumulate(api.Scene(flat), small.render_desc(accel=accel))
    print("build %d: %.1f ms, depth %d" % (rep, st["tree_build_ms"], st["tree_depth"]), flush=True)
