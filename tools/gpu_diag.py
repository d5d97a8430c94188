import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustraytracer_amd import api, config
from oracle import oracle_py as O
be = api.Backend(0)
for name in ("cornell_box", "smallpt"):
    flat = config.load_scene(os.path.join(ROOT, "configs", name, "scene.toml"))
    sc = api.Scene(flat)
    rng = np.random.default_rng(7)
    u, v = rng.random(400000), rng.random(400000)
    ref = O.trace_primary(flat, u, v)
    oid, pid, t = be.trace_primary(sc, u, v, precision=api.PRECISION_F32)
    same = (oid == ref[0]) & (pid == ref[1]) & (ref[0] >= 0)
    rel = np.abs(t[same] - ref[2][same]) / ref[2][same]
    print(name, "mismatch", int(np.sum((oid != ref[0]) | (pid != ref[1]))), "max rel", rel.max(), "p99.9", np.quantile(rel, 0.999))
    for o in sorted(set(ref[0][same].tolist())):
        m = ref[0][same] == o
        print("   obj", o, "n", int(m.sum()), "max rel %.3e" % rel[m].max(), "median %.3e" % np.median(rel[m]))
    bad = np.nonzero((oid != ref[0]) | (pid != ref[1]))[0]
    for i in bad[:5]:
        print("   mismatch at", u[i], v[i], "gpu", oid[i], pid[i], t[i], "ref", ref[0][i], ref[1][i], ref[2][i])
