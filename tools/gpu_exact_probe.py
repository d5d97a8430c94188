import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import scenes_util as S
from oracle import oracle_py as O
from rustraytracer_amd import api, config
be = api.Backend(0)
scenes = {n: config.load_scene("/root/repo/configs/%s/scene.toml" % n) for n in ("cornell_box", "smallpt")}
scenes["all_prims"] = S.all_prims_scene()
g = np.load("/root/repo/tests/golden/primary_hits.npz")
for name in ("cornell_box", "smallpt"):
    sc = api.Scene(scenes[name])
    got = be.trace_primary(sc, g[name + "_u"], g[name + "_v"], precision=api.PRECISION_F32)
    bad = (got[0] != g[name + "_object_id"]) | (got[1] != g[name + "_prim_id"])
    print("golden", name, "bad", int(bad.sum()), "of", len(bad))
rng = np.random.default_rng(99)
u, v = rng.random(300000), rng.random(300000)
for name, flat in scenes.items():
    sc = api.Scene(flat)
    ref = O.trace_primary(flat, u, v)
    got = be.trace_primary(sc, u, v, precision=api.PRECISION_F32)
    bad = (got[0] != ref[0]) | (got[1] != ref[1])
    ok = ~bad & (ref[0] >= 0)
    rel = np.abs(got[2][ok] - ref[2][ok]) / ref[2][ok]
    print("primary", name, "bad", int(bad.sum()), "max rel t", rel.max())
    o, d = S.secondary_rays(flat, O, 100000, seed=17)
    ref = O.intersect(flat, o, d)
    got = be.intersect(sc, o, d, precision=api.PRECISION_F32)
    bad = (got[0] != ref[0]) | (got[1] != ref[1])
    with np.errstate(invalid="ignore", divide="ignore"):
        tie = bad & (np.abs(got[2] - ref[2]) <= 1e-5 * np.abs(ref[2]))
    g64 = be.intersect(sc, o, d, precision=api.PRECISION_F64)
    bad64 = (g64[0] != ref[0]) | (g64[1] != ref[1])
    same_as_64 = (got[0] == g64[0]) & (got[1] == g64[1])
    print("secondary", name, "bad", int(bad.sum()), "non-tie", int((bad & ~tie).sum()), "f64 kernel bad", int(bad64.sum()), "f32 != f64 kernel", int((~same_as_64).sum()))
    for tree in (api.ACCEL_SAH_GPU, api.ACCEL_LBVH):
        gt = be.intersect(sc, o, d, precision=api.PRECISION_F32, accel=tree)
        print("   tree", tree, "vs scan same:", all(np.array_equal(a, b) for a, b in zip(gt, got)), "ids differ", int(((gt[0] != got[0]) | (gt[1] != got[1])).sum()), "t differ", int((gt[2] != got[2]).sum()))
# render with the flag
rc = config.stratified_config(96, 72, 8, 8, seed=41)
for name in ("cornell_box", "smallpt"):
    sc = api.Scene(scenes[name])
    ref, rst = O.render_accumulate(scenes[name], rc.render_desc(), device_numbers="f32")
    for flags in (0, api.RENDER_EXACT_HITS):
        dev, st = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F32, pipeline=api.PIPELINE_FUSED, flags=flags))
        err = np.abs(dev.astype(np.float64) - ref)
        off = np.any(err > 2e-5 * np.abs(ref) + 1e-6, axis=-1)
        print("render", name, "flags", flags, "pixels off", int(off.sum()), "rays", st["rays"], "oracle", rst["rays_live"], "redos", st["exact_redos"], "ms", st["kernel_ms"])
