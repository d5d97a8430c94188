#!/usr/bin/env python3
"""Regenerate the Cornell-box PLY meshes and assets/test.ply.

The reference repository stores its eight .ply files as git-LFS pointer stubs
(three text lines: version / oid / size), so the geometry itself is not
available offline.  The stub sizes pin the layout: every file is a binary PLY
with 24-byte vertices (x, y, z, nx, ny, nz as float32) and 17-byte faces
(uchar count 4 + four uint32 indices), i.e. 18 quads in total and no triangle:

    409 B  -> 4 vertices, 1 quad   (light, red wall, green wall)
    636 B  -> 12 vertices, 3 quads (floor + ceiling + back wall)
    975 B  -> 24 vertices, 6 quads (large box, small box, assets/test.ply)

The meshes written here are RECONSTRUCTIONS from the canonical Cornell-box
measurements (millimetres, y up) combined with the transforms stored in
configs/cornell_box/scene.toml, which map PLY (x, y, z) to world
(-s*x, s*z, s*y) + t.  They have the originals' byte sizes but are not byte-identical to them.

  * room: x in [0, 556], y in [0, 548.8], z in [0, 559.2]; the opening faces -z
  * light: 130 x 105 mm quad centred on its local origin (the scene transform
    translates it to 0.8 mm below the ceiling)
  * large box: unit cube (half extent 0.5), same file as assets/test.ply
    (the two LFS stubs carry the same sha256)
  * small box: side-2 cube (half extent 1)

Faces are wound so that (v1-v0) x (v2-v0) points into the room for walls and
out of the solid for boxes, because the reference never flips a normal toward
the incoming ray (src/material.rs:67-91 scatters about the stored normal).

The property set is exactly what the reference loader demands
(src/shapes/utils.rs:24-69): float x,y,z,nx,ny,nz and
`list uchar uint vertex_indices`.
"""
import os
import struct
import sys

import numpy as np


def write_ply(path, quads, comment):
    """quads: list of 4x3 arrays; one vertex per face corner, flat normals."""
    verts = []
    faces = []
    for q in quads:
        q = np.asarray(q, dtype=np.float64)
        n = np.cross(q[1] - q[0], q[2] - q[0])
        n = n / np.linalg.norm(n)
        base = len(verts)
        for p in q:
            verts.append((p[0], p[1], p[2], n[0], n[1], n[2]))
        faces.append((base, base + 1, base + 2, base + 3))
    header = (
        "ply\n"
        "format binary_little_endian 1.0\n"
        f"comment {comment}\n"
        f"element vertex {len(verts)}\n"
        "property float x\nproperty float y\nproperty float z\n"
        "property float nx\nproperty float ny\nproperty float nz\n"
        f"element face {len(faces)}\n"
        "property list uchar uint vertex_indices\n"
        "end_header\n"
    )
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        for v in verts:
            f.write(struct.pack("<6f", *v))
        for fc in faces:
            f.write(struct.pack("<B4I", 4, *fc))


def cube_quads(h):
    """Axis-aligned cube of half extent h, outward-facing quads."""
    return [
        # +x
        [(h, -h, -h), (h, h, -h), (h, h, h), (h, -h, h)],
        # -x
        [(-h, -h, -h), (-h, -h, h), (-h, h, h), (-h, h, -h)],
        # +y
        [(-h, h, -h), (-h, h, h), (h, h, h), (h, h, -h)],
        # -y
        [(-h, -h, -h), (h, -h, -h), (h, -h, h), (-h, -h, h)],
        # +z
        [(-h, -h, h), (h, -h, h), (h, h, h), (-h, h, h)],
        # -z
        [(-h, -h, -h), (-h, h, -h), (h, h, -h), (h, -h, -h)],
    ]


def main(root):
    X, Y, Z = 556.0, 548.8, 559.2
    # 63 characters, which makes every file exactly as long as its LFS stub says
    note = "reconstructed RustRayTracer LFS mesh: tools/make_cornell_ply.py"
    cb = os.path.join(root, "configs", "cornell_box")
    os.makedirs(cb, exist_ok=True)
    os.makedirs(os.path.join(root, "assets"), exist_ok=True)

    floor = [(0, 0, 0), (0, 0, Z), (X, 0, Z), (X, 0, 0)]
    ceiling = [(0, Y, 0), (X, Y, 0), (X, Y, Z), (0, Y, Z)]
    back = [(0, 0, Z), (0, Y, Z), (X, Y, Z), (X, 0, Z)]
    green = [(0, 0, 0), (0, Y, 0), (0, Y, Z), (0, 0, Z)]
    red = [(X, 0, 0), (X, 0, Z), (X, Y, Z), (X, Y, 0)]
    light = [(-65, 0, -52.5), (65, 0, -52.5), (65, 0, 52.5), (-65, 0, 52.5)]

    write_ply(os.path.join(cb, "cornell_box_0000_m000.ply"), [floor, ceiling, back], note)
    write_ply(os.path.join(cb, "cornell_box_0000_m001.ply"), [red], note)
    write_ply(os.path.join(cb, "cornell_box_0000_m002.ply"), [green], note)
    write_ply(os.path.join(cb, "Light_0000_m000.ply"), [light], note)
    write_ply(os.path.join(cb, "large_box_0000_m000.ply"), cube_quads(0.5), note)
    write_ply(os.path.join(cb, "small_box_0000_m000.ply"), cube_quads(1.0), note)
    write_ply(os.path.join(root, "assets", "test.ply"), cube_quads(0.5), note)
    write_ply(os.path.join(root, "assets", "large_box_0000_m000.ply"), cube_quads(0.5), note)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), ".."))
