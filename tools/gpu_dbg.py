import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rustraytracer_amd import api, config
from oracle import oracle_py as O
import scenes_util as S
be = api.Backend(0)
flat = config.load_scene(os.path.join(ROOT, "configs", "cornell_box", "scene.toml"))
sc = api.Scene(flat)
o, d = S.secondary_rays(flat, O, 40000, seed=23)
for prec in (1, 0):
    lin = be.intersect(sc, o, d, precision=prec, accel=api.ACCEL_LINEAR)
    bvh = be.intersect(sc, o, d, precision=prec, accel=api.ACCEL_LBVH)
    idbad = (lin[0] != bvh[0]) | (lin[1] != bvh[1])
    tbad = ~idbad & (lin[2] != bvh[2])
    print("prec", prec, "id mismatches", int(idbad.sum()), "t-only mismatches", int(tbad.sum()))
    for i in np.nonzero(idbad)[0][:6]:
        print("   id", i, "lin", lin[0][i], lin[1][i], repr(lin[2][i]), "bvh", bvh[0][i], bvh[1][i], repr(bvh[2][i]), "o", o[i], "d", d[i])
    for i in np.nonzero(tbad)[0][:6]:
        print("   t ", i, "lin", lin[0][i], lin[1][i], repr(lin[2][i]), "bvh", repr(bvh[2][i]), "rel", abs(lin[2][i]-bvh[2][i])/lin[2][i])
