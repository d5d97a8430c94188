"""Where the tree starts to beat the linear scan, and the wavefront pipeline the fused kernel: triangle meshes of growing size.
   gpu_crossover.py [small|large]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rustraytracer_amd import api, config
import test_gpu_lbvh as T
be = api.Backend(0)
large = len(sys.argv) > 1 and sys.argv[1] == "large"
rc = config.stratified_config(1024, 1024, 16, 16)
sizes = (6000, 24000, 100000, 400000) if large else (24, 48, 96, 192, 384, 768, 1536, 6000)
for n in sizes:
    flat = T.mesh_scene(n)
    sc = api.Scene(flat)
    row = []
    combos = ((api.ACCEL_LINEAR, api.PIPELINE_FUSED), (api.ACCEL_LBVH, api.PIPELINE_FUSED), (api.ACCEL_LBVH, api.PIPELINE_WAVEFRONT),
              (api.ACCEL_SAH, api.PIPELINE_FUSED), (api.ACCEL_SAH, api.PIPELINE_WAVEFRONT))
    for accel, pipe in combos:
        if large and accel == api.ACCEL_LINEAR:
            row.append(float("nan"))
            continue
        for rep in range(2):
            rgb, st = be.render_accumulate(sc, rc.render_desc(accel=accel, pipeline=pipe))
        row.append(st["rays"] / st["kernel_ms"] / 1e3)
    print("prims %7d  linear/fused %8.0f  cluster tree fused %8.0f wavefront %8.0f  surface-area tree fused %8.0f wavefront %8.0f Mrays/s"
          % (flat.n_prims, *row), flush=True)
if large:
    import make_c5_assets
    make_c5_assets.main()
    flat = config.load_scene(os.path.join(ROOT, "configs/mesh_1m/scene.toml"))
    sc = api.Scene(flat)
    rc = config.stratified_config(1920, 1080, 8, 8)
    row = []
    for accel in (api.ACCEL_LBVH, api.ACCEL_SAH):
        for pipe in (api.PIPELINE_FUSED, api.PIPELINE_WAVEFRONT):
            for rep in range(2):
                rgb, st = be.render_accumulate(sc, rc.render_desc(accel=accel, pipeline=pipe))
            row.append(st["rays"] / st["kernel_ms"] / 1e3)
    print("mesh_1m %7d prims  cluster tree fused %8.0f wavefront %8.0f  surface-area tree fused %8.0f wavefront %8.0f Mrays/s"
          % (flat.n_prims, *row), flush=True)
