"""Writes the f32 boxes of a scene's bounded primitives (6 floats each: lo, hi) for tools/tree_cost.cpp:
   export_boxes.py SCENE.toml OUT.f32      (configs/mesh_1m/scene.toml needs tools/make_c5_assets.py first)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustraytracer_amd import _abi, config
flat = config.load_scene(sys.argv[1])
p = flat.prims
keep = (p["type"] != _abi.PRIM_PLANE)
v = p["v"][:, :12].reshape(-1, 4, 3)
lo, hi = v[:, :3].min(axis=1), v[:, :3].max(axis=1)
quad = p["type"] == _abi.PRIM_QUAD
lo[quad], hi[quad] = np.minimum(lo[quad], v[quad, 3]), np.maximum(hi[quad], v[quad, 3])
sph = p["type"] == _abi.PRIM_SPHERE
lo[sph], hi[sph] = v[sph, 0] - np.abs(p["v"][sph, 3:4]), v[sph, 0] + np.abs(p["v"][sph, 3:4])
np.concatenate([lo[keep], hi[keep]], axis=1).astype(np.float32).tofile(sys.argv[2])
print(int(keep.sum()), "boxes ->", sys.argv[2])
