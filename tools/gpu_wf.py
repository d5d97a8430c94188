"""Wavefront pipeline vs fused pipeline: statistics, images, throughput."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rustraytracer_amd import api, config
be = api.Backend(0)
def run(name, flat, rc, accel, reps=1):
    sc = api.Scene(flat)
    out = {}
    for pipe, pname in ((api.PIPELINE_FUSED, "fused"), (api.PIPELINE_WAVEFRONT, "wavefront")):
        for rep in range(reps):
            rgb, st = be.render_accumulate(sc, rc.render_desc(accel=accel, pipeline=pipe))
        out[pname] = (rgb, st)
        print(name, pname, {k: st[k] for k in ("samples", "rays", "prim_tests", "node_tests", "nonfinite")}, "bounces", st["bounces"],
              "ms %.2f Msamples/s %.1f Mrays/s %.1f" % (st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3, st["rays"] / st["kernel_ms"] / 1e3), flush=True)
    a, b = out["fused"][0], out["wavefront"][0]
    print(name, "image mean fused %.6f wavefront %.6f max |diff| / spp %.3g  rel L1 %.3g" % (a.mean() / rc.spp, b.mean() / rc.spp,
          np.abs(a - b).max() / rc.spp, np.abs(a - b).sum() / np.abs(a).sum()), flush=True)
cornell = config.load_scene(os.path.join(ROOT, "configs/cornell_box/scene.toml"))
smallpt = config.load_scene(os.path.join(ROOT, "configs/smallpt/scene.toml"))
run("cornell 64x64@16", cornell, config.stratified_config(64, 64, 4, 4), api.ACCEL_LINEAR)
run("cornell 37x21@9 random", cornell, config.random_config(37, 21, 9), api.ACCEL_LINEAR)
run("smallpt 200x150@64", smallpt, config.stratified_config(200, 150, 8, 8), api.ACCEL_LINEAR)
run("cornell 500x500@256", cornell, config.stratified_config(500, 500, 16, 16), api.ACCEL_LINEAR, reps=2)
run("cornell 500x500@256 lbvh", cornell, config.stratified_config(500, 500, 16, 16), api.ACCEL_LBVH, reps=2)
if len(sys.argv) > 1:
    import make_c5_assets
    make_c5_assets.main()
    c5 = config.load_scene(os.path.join(ROOT, "configs/mesh_1m/scene.toml"))
    run("mesh_1m 960x540@64", c5, config.stratified_config(960, 540, 8, 8), api.ACCEL_LBVH, reps=2)
