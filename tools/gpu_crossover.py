"""Where the tree starts to beat the linear scan: triangle meshes of growing size, three configurations."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rustraytracer_amd import api, config
import test_gpu_lbvh as T
be = api.Backend(0)
rc = config.stratified_config(1024, 1024, 16, 16)
for n in (24, 48, 96, 192, 384, 768, 1536, 6000):
    flat = T.mesh_scene(n)
    sc = api.Scene(flat)
    row = []
    for accel, pipe in ((api.ACCEL_LINEAR, api.PIPELINE_FUSED), (api.ACCEL_LBVH, api.PIPELINE_FUSED), (api.ACCEL_LBVH, api.PIPELINE_WAVEFRONT)):
        for rep in range(2):
            rgb, st = be.render_accumulate(sc, rc.render_desc(accel=accel, pipeline=pipe))
        row.append(st["rays"] / st["kernel_ms"] / 1e3)
    print("prims %5d  linear/fused %8.0f  tree/fused %8.0f  tree/wavefront %8.0f Mrays/s" % (flat.n_prims, *row), flush=True)
