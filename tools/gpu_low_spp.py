"""Both pipelines at 1 .. 256 samples per pixel (device only): kernel time of the Cornell box and smallpt at 500x500 and
1920x1080.  (With work items of one or two samples the fused kernel waits for its work-item counter -- one atomic per 64 items --
which is why items are never shorter than eight samples: 500x500 @ 4 spp 0.32 -> 0.15 ms.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustraytracer_amd import api, config
be = api.Backend(0)
for scene_name in ("cornell_box", "smallpt"):
    sc = api.Scene(config.load_scene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", scene_name, "scene.toml")))
    for w, h in ((500, 500), (1920, 1080)):
        for side in (1, 2, 4, 8, 16):
            rc = config.stratified_config(w, h, side, side, seed=1)
            t = {}
            for name, pl in (("fused", api.PIPELINE_FUSED), ("wavefront", api.PIPELINE_WAVEFRONT)):
                be.render_accumulate(sc, rc.render_desc(pipeline=pl))
                best = 1e9
                for _ in range(3):
                    _, st = be.render_accumulate(sc, rc.render_desc(pipeline=pl))
                    best = min(best, st["kernel_ms"])
                t[name] = best
            print("%-12s %4dx%-4d @ %3d spp: fused %.3f ms, wavefront %.3f ms" % (scene_name, w, h, side * side, t["fused"], t["wavefront"]), flush=True)
