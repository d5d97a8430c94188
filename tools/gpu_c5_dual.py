"""Do two wavefront renders of the million-triangle scene overlap on one GPU?  Two backends (two streams, two sets of
queues) each trace half of the sample indices from two host threads; the pair is timed against one backend tracing all."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_c5_assets
make_c5_assets.main()
from rustraytracer_amd import api, config
W, H, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 2
bes = [api.Backend(0) for _ in range(N)]
flat = config.load_scene(os.path.join(ROOT, "configs/mesh_1m/scene.toml"))
sc = api.Scene(flat)
rc = config.stratified_config(W, H, S, S)
small = config.stratified_config(64, 64, 2, 2)
for be in bes:
    be.render_accumulate(sc, small.render_desc(accel=api.ACCEL_LBVH))
spp = rc.spp
for rep in range(2):
    t0 = time.time(); _, st = bes[0].render_accumulate(sc, rc.render_desc(accel=api.ACCEL_LBVH)); t1 = time.time() - t0
    print("one backend : wall %.1f ms, kernel %.1f ms, %.1f Mrays/s (wall)" % (t1 * 1e3, st["kernel_ms"], st["rays"] / t1 / 1e6), flush=True)
    out = [None] * N
    def work(i):
        out[i] = bes[i].render_accumulate(sc, rc.render_desc(sample_begin=i * spp // N, sample_end=(i + 1) * spp // N, accel=api.ACCEL_LBVH))
    th = [threading.Thread(target=work, args=(i,)) for i in range(N)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    t2 = time.time() - t0
    rays = sum(o[1]["rays"] for o in out)
    print("%d backends  : wall %.1f ms, kernels %s ms, %.1f Mrays/s (wall)" % (N, t2 * 1e3, [round(o[1]["kernel_ms"], 1) for o in out], rays / t2 / 1e6), flush=True)
