#!/usr/bin/env python3
"""Procedural triangle meshes in the PLY dialect the reference loader accepts
(src/shapes/utils.rs:24-69: float x,y,z,nx,ny,nz + `list uchar uint vertex_indices`, binary little endian).

    python tools/make_mesh.py OUT.ply --triangles 1000000 [--radius 120 --bump 0.08 --seed 1]

The surface is a sphere displaced by a few low-frequency lobes (deterministic in --seed), tessellated as a
latitude/longitude grid; every grid cell gives two triangles wound counter-clockwise seen from outside, so the
face normal (v1-v0) x (v2-v0) the reference shades with points outward.  The pole rows are left out (no
degenerate triangles).  Mind the reference's ABSOLUTE determinant threshold (|e1 . (d x e2)| < 1e-6 rejects,
src/shapes/triangle.rs:36): triangles must keep an area well above 1e-6 in scene units, so a million-triangle
mesh belongs in a scene measured in millimetres, not metres.
"""
import argparse
import struct

import numpy as np


def displaced_sphere(n_triangles, radius=1.0, bump=0.08, seed=1):
    """-> (vertices [n,3] f32, normals [n,3] f32, faces [m,3] u32) with m <= n_triangles, m close to it."""
    n_lon = max(3, int(np.sqrt(n_triangles / 2.0 * 2.0)))
    n_lat = max(2, n_triangles // (2 * n_lon))
    rng = np.random.default_rng(seed)
    theta = np.linspace(0.02 * np.pi, 0.98 * np.pi, n_lat + 1)
    phi = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    T, P = np.meshgrid(theta, phi, indexing="ij")
    r = np.ones_like(T)
    for _ in range(6):
        a, b = rng.integers(1, 6, size=2)
        ph0, th0 = rng.random(2) * 2 * np.pi
        r += bump / 3 * np.sin(a * T + th0) * np.cos(b * P + ph0)
    r *= radius
    x = r * np.sin(T) * np.cos(P)
    y = r * np.cos(T)
    z = r * np.sin(T) * np.sin(P)
    v = np.stack([x, y, z], -1).reshape(-1, 3)
    nrm = v / np.linalg.norm(v, axis=1, keepdims=True)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = (i * n_lon + j).ravel()
    b = (i * n_lon + (j + 1) % n_lon).ravel()
    c = ((i + 1) * n_lon + j).ravel()
    d = ((i + 1) * n_lon + (j + 1) % n_lon).ravel()
    faces = np.concatenate([np.stack([a, b, d], 1), np.stack([a, d, c], 1)]).astype(np.uint32)
    # outward winding check on one face
    f = faces[0]
    if np.dot(np.cross(v[f[1]] - v[f[0]], v[f[2]] - v[f[0]]), v[f[0]]) < 0:
        faces = faces[:, ::-1].copy()
    return v.astype(np.float32), nrm.astype(np.float32), faces


def write_ply(path, v, nrm, faces, comment="procedural mesh, tools/make_mesh.py"):
    header = ("ply\nformat binary_little_endian 1.0\ncomment %s\nelement vertex %d\n" % (comment, len(v))
              + "".join(f"property float {k}\n" for k in ("x", "y", "z", "nx", "ny", "nz"))
              + "element face %d\nproperty list uchar uint vertex_indices\nend_header\n" % len(faces))
    rec = np.empty(len(faces), dtype=[("n", "u1"), ("i", "<u4", (faces.shape[1],))])
    rec["n"] = faces.shape[1]
    rec["i"] = faces
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.concatenate([v, nrm], axis=1).astype("<f4").tobytes())
        f.write(rec.tobytes())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--triangles", type=int, default=1000000)
    ap.add_argument("--radius", type=float, default=1.0)
    ap.add_argument("--bump", type=float, default=0.08)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    v, n, f = displaced_sphere(a.triangles, a.radius, a.bump, a.seed)
    write_ply(a.out, v, n, f)
    print(f"{a.out}: {len(v)} vertices, {len(f)} triangles")
