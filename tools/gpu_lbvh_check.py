import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rustraytracer_amd import api, config, _abi
from oracle import oracle_py as O
import scenes_util as S
import make_mesh
be = api.Backend(0)
def check(name, flat, n=50000, oracle=True):
    sc = api.Scene(flat)
    o, d = S.secondary_rays(flat, O, n, seed=3) if oracle else (None, None)
    if o is None:
        rng = np.random.default_rng(1); u, v = rng.random(n), rng.random(n)
        a = be.trace_primary(sc, u, v, precision=api.PRECISION_F32, accel=api.ACCEL_LINEAR)
        t0 = time.time(); b = be.trace_primary(sc, u, v, precision=api.PRECISION_F32, accel=api.ACCEL_LBVH); dt = time.time() - t0
        print(name, "prims", flat.n_prims, "lbvh vs linear (f32, primary) mismatches", int(np.sum((a[0] != b[0]) | (a[1] != b[1]))), "max |dt|", np.nanmax(np.abs(np.where(np.isinf(a[2]), 0, a[2] - b[2]))), "lbvh call %.3fs" % dt)
        return
    ref = O.intersect(flat, o, d)
    for prec in (api.PRECISION_F64, api.PRECISION_F32):
        lin = be.intersect(sc, o, d, precision=prec, accel=api.ACCEL_LINEAR)
        bvh = be.intersect(sc, o, d, precision=prec, accel=api.ACCEL_LBVH)
        print(name, "prims", flat.n_prims, "prec", prec, "lbvh!=linear", int(np.sum((lin[0] != bvh[0]) | (lin[1] != bvh[1]))),
              "lbvh!=oracle", int(np.sum((ref[0] != bvh[0]) | (ref[1] != bvh[1]))), "t equal", bool(np.array_equal(lin[2], bvh[2])))
for name in ("cornell_box", "smallpt"):
    check(name, config.load_scene(os.path.join(ROOT, "configs", name, "scene.toml")))
check("all_prims", S.all_prims_scene())
v, nrm, f = make_mesh.displaced_sphere(4000, radius=1.2, bump=0.2, seed=2)
def mesh_scene(v, f, extra=True):
    prims = [(2, 0, i, v[fc].astype(np.float64) + np.array([0, 0, -4.0])) for i, fc in enumerate(f)]
    mats = [(_abi.MAT_LAMBERTIAN, (0.7, 0.7, 0.7), 0.0), (_abi.MAT_EMISSIVE, (1, 1, 1), 0.0), (_abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)]
    prims.append((0, 1, 0, [0, 0, 0, 40.0]))
    prims.append((1, 2, 0, [0, -1.5, 0, 0, 1, 0]))
    return S.make_scene(prims, mats, vfov=50.0)
check("mesh4k", mesh_scene(v, f), n=20000)
v, nrm, f = make_mesh.displaced_sphere(1000000, radius=1.2, bump=0.2, seed=2)
t0 = time.time(); flat = mesh_scene(v, f); print("1M scene flatten %.1fs" % (time.time() - t0))
t0 = time.time(); sc = api.Scene(flat); print("scene create %.2fs" % (time.time() - t0))
rng = np.random.default_rng(1); u, vv = rng.random(200000), rng.random(200000)
t0 = time.time(); b = be.trace_primary(sc, u, vv, accel=api.ACCEL_LBVH); print("first lbvh call (build) %.2fs" % (time.time() - t0))
t0 = time.time(); b = be.trace_primary(sc, u, vv, accel=api.ACCEL_LBVH); print("second lbvh call %.3fs, hits on mesh %d" % (time.time() - t0, int((b[0] == 0).sum())))
sub = slice(0, 2000)
a = be.trace_primary(sc, u[sub], vv[sub], accel=api.ACCEL_LINEAR)
print("1M: lbvh vs linear mismatches on 2000 rays:", int(np.sum((a[0] != b[0][sub]) | (a[1] != b[1][sub]))))
rc = config.stratified_config(256, 256, 8, 8)
rgb, st = be.render_accumulate(sc, rc.render_desc(accel=api.ACCEL_LBVH))
print("render 1M:", st, "Msamples/s %.1f Mrays/s %.1f" % (st["samples"] / st["kernel_ms"] / 1e3, st["rays"] / st["kernel_ms"] / 1e3))
