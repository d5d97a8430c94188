import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rustraytracer_amd import api, config
be = api.Backend(0)
flat = config.load_scene(os.path.join(ROOT, "configs/cornell_box/scene.toml"))
sc = api.Scene(flat)
rng = np.random.default_rng(1); u, v = rng.random(1000), rng.random(1000)
for prec in (api.PRECISION_F32, api.PRECISION_F64):
    a = be.trace_primary(sc, u, v, precision=prec, accel=api.ACCEL_LINEAR)
    b = be.trace_primary(sc, u, v, precision=prec, accel=api.ACCEL_LBVH)
    print("prec", prec, "mismatch", int(np.sum(a[0] != b[0])), "lbvh hits", int(np.sum(b[0] >= 0)), "linear hits", int(np.sum(a[0] >= 0)))
rc = config.stratified_config(64, 64, 2, 2)
rgb, st = be.render_accumulate(sc, rc.render_desc(accel=api.ACCEL_LBVH, pipeline=api.PIPELINE_FUSED))
print(st)
