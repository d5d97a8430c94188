#!/usr/bin/env python3
"""Generate the two 500k-triangle meshes of configs/mesh_1m (deterministic; ~25 MB, not committed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_mesh  # noqa: E402


def main():
    out = os.path.join(ROOT, "assets", "generated")
    os.makedirs(out, exist_ok=True)
    for name, radius, seed in (("blob_a_500k.ply", 140.0, 11), ("blob_b_500k.ply", 100.0, 12)):
        path = os.path.join(out, name)
        if not os.path.exists(path):
            v, n, f = make_mesh.displaced_sphere(500000, radius=radius, bump=0.25, seed=seed)
            make_mesh.write_ply(path, v, n, f)
    return out


if __name__ == "__main__":
    print(main())
