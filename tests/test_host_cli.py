"""The C++ command-line host (rustraytracer_amd/host): config parsing, PLY loading, scene flattening
and image writing, checked against the Python mirror of the same reference code and against the
reference's own fixtures.  No GPU: rendering itself is covered by the GPU tests."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from rustraytracer_amd import _abi, config

# RRT_HOST_BINARY: another build of the host to run these tests on (the sanitizer build, see the last test)
RRT = os.path.abspath(os.environ.get("RRT_HOST_BINARY", os.path.join(ROOT, "rustraytracer_amd", "host", "rrt")))


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(RRT):
        import __graft_entry__ as g
        g.build()
    assert os.path.exists(RRT)


def run(*args, cwd=ROOT):
    return subprocess.run([RRT, *args], cwd=cwd, capture_output=True, text=True, timeout=120)


def read_dump(path):
    with open(path, "rb") as f:
        magic, n_prims, n_objects, _ = struct.unpack("<4I", f.read(16))
        assert magic == 0x53545252
        cam = np.frombuffer(f.read(96), dtype=np.float64).reshape(4, 3)
        prims = np.frombuffer(f.read(n_prims * _abi.PRIM_DTYPE.itemsize), dtype=_abi.PRIM_DTYPE)
        mats = np.frombuffer(f.read(n_objects * _abi.MATERIAL_DTYPE.itemsize), dtype=_abi.MATERIAL_DTYPE)
        assert f.read() == b""
    return cam, prims, mats


def same_scene(dump, flat):
    cam, prims, mats = dump
    for k, name in enumerate(("origin", "lower_left_corner", "horizontal", "vertical")):
        assert np.allclose(cam[k], flat.camera[name], rtol=1e-14, atol=1e-15)
    assert len(prims) == flat.n_prims and len(mats) == flat.n_objects
    for k in ("type", "object_id", "prim_id"):
        assert np.array_equal(prims[k], flat.prims[k])
    # numpy and the C++ host sum the matrix products in different orders
    assert np.allclose(prims["v"], flat.prims["v"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(mats["type"], flat.materials["type"])
    assert np.array_equal(mats["rgb"], flat.materials["rgb"]) and np.array_equal(mats["param"], flat.materials["param"])


@pytest.mark.parametrize("name", ["cornell_box", "smallpt", "cornell_cube"])
def test_flattened_scene_equals_python_mirror(tmp_path, name):
    scene = os.path.join(ROOT, "configs", name, "scene.toml")
    out = str(tmp_path / "scene.bin")
    r = run("-s", scene, "--dump-scene", out)
    assert r.returncode == 0, r.stderr
    same_scene(read_dump(out), config.load_scene(scene))


def test_reference_fixture_test_ply(tmp_path):
    """src/shapes/utils.rs:126-133: assets/test.ply has 24 vertices / 24 normals / 6 faces."""
    scene = tmp_path / "s.toml"
    scene.write_text('''
[camera]
type = "Perspective"
look_from = { x = 0.0, y = 0.0, z = 5.0 }
look_at = { x = 0.0, y = 0.0, z = 0.0 }
vup = { x = 0.0, y = 1.0, z = 0.0 }
vfov = 40.0
aspect = 1.0
[[objects]]
[objects.shape]
type = "Mesh"
file = "assets/test.ply"
[objects.material]
type = "IdealReflector"
''')
    out = str(tmp_path / "scene.bin")
    r = run("-s", str(scene), "--dump-scene", out)
    assert r.returncode == 0, r.stderr
    cam, prims, mats = read_dump(out)
    assert len(prims) == 6 and np.all(prims["type"] == _abi.PRIM_QUAD) and list(prims["prim_id"]) == list(range(6))
    assert mats["type"][0] == _abi.MAT_REFLECTOR


SCENE_HEAD = '''
[camera]
type = "Perspective"
look_from = { x = 0, y = 0, z = 5 }   # integers where the schema says f64: serde accepts them
look_at = { x = 0.0, y = 0.0, z = 0.0 }
vup = { x = 0.0, y = 1.0, z = 0.0 }
vfov = 4_0.0
aspect = 1e0
'''


def write_ply(path, fmt, vtype="float", itype="uint", ctype="uchar", faces=((0, 1, 2), (0, 2, 3)), extra_prop=False):
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float64)
    n = np.array([[0, 0, 1]] * 4, dtype=np.float64)
    npt = {"float": "f4", "double": "f8", "uint": "u4", "int": "i4", "uchar": "u1", "ushort": "u2"}
    e = {"ascii": "", "binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
    head = ["ply", f"format {fmt} 1.0", "comment made by a test", "element vertex 4"]
    head += [f"property {vtype} {k}" for k in ("x", "y", "z", "nx", "ny", "nz")]
    if extra_prop:
        head.append("property uchar red")
    head += [f"element face {len(faces)}", f"property list {ctype} {itype} vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        for i in range(4):
            row = list(v[i]) + list(n[i])
            if fmt == "ascii":
                f.write((" ".join(repr(float(x)) for x in row) + (" 200" if extra_prop else "") + "\n").encode())
            else:
                f.write(np.array(row, dtype=e + npt[vtype]).tobytes())
                if extra_prop:
                    f.write(b"\xc8")
        for fc in faces:
            if fmt == "ascii":
                f.write((" ".join(str(x) for x in (len(fc), *fc)) + "\n").encode())
            else:
                f.write(np.array([len(fc)], dtype=e + npt[ctype]).tobytes())
                f.write(np.array(fc, dtype=e + npt[itype]).tobytes())


def mesh_scene(tmp_path, ply_name, transform=""):
    s = tmp_path / "s.toml"
    s.write_text(SCENE_HEAD + f'''
[[objects]]
[objects.shape]
type = "Mesh"
file = "{ply_name}"
{transform}
[objects.material]
type = "Lambertian"
albedo = {{ x = 0.5, y = 0.5, z = 0.5 }}
''')
    return str(s)


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_formats_and_mixed_faces(tmp_path, fmt):
    ply = tmp_path / "m.ply"
    write_ply(str(ply), fmt, faces=((0, 1, 2), (0, 1, 2, 3)), ctype="ushort", extra_prop=True)
    tr = ("transform = { m11 = 2.0, m12 = 0.0, m13 = 0.0, m14 = 0.0, m21 = 0.0, m22 = 1.0, m23 = 0.0, m24 = 0.0, "
          "m31 = 0.0, m32 = 0.0, m33 = 1.0, m34 = 0.0, m41 = 0.5, m42 = -1.0, m43 = 0.25, m44 = 2.0 }")
    scene = mesh_scene(tmp_path, str(ply), tr)
    out = str(tmp_path / "scene.bin")
    r = run("-s", scene, "--dump-scene", out)
    assert r.returncode == 0, r.stderr
    dump = read_dump(out)
    same_scene(dump, config.load_scene(scene))
    assert list(dump[1]["type"]) == [_abi.PRIM_TRIANGLE, _abi.PRIM_QUAD]
    # column-major matrix with w = 2: (2x + 0.5, y - 1, z + 0.25) / 2
    assert np.allclose(dump[1]["v"][1, :12].reshape(4, 3)[2], [(2 * 1 + 0.5) / 2, (1 - 1) / 2, 0.25 / 2])


@pytest.mark.parametrize("kw,msg", [
    (dict(vtype="double"), "x's type unrecognized"),          # src/shapes/utils.rs:28-31
    (dict(itype="int"), "vertex_indices's type unrecognized"),  # src/shapes/utils.rs:61-64
    (dict(faces=((0, 1),)), "Invalid mesh"),                   # src/shapes/utils.rs:92-94
    (dict(faces=((0, 1, 2, 3, 0),)), "non-triangle or non-quadrilateral"),  # src/shapes/mesh.rs:75
    (dict(faces=((0, 1, 2, 9),)), "out of range"),
])
def test_ply_strictness(tmp_path, kw, msg):
    ply = tmp_path / "m.ply"
    write_ply(str(ply), "binary_little_endian", **kw)
    scene = mesh_scene(tmp_path, str(ply))
    r = run("-s", scene, "--dump-scene", str(tmp_path / "o.bin"))
    assert r.returncode == 101 and msg in r.stderr
    with pytest.raises(config.ConfigError):
        config.load_scene(scene)


def test_quad_validation(tmp_path):
    """src/shapes/utils.rs:98-116: a quad must be coplanar and convex."""
    for verts, reason in (([[0, 0, 0], [1, 0, 0], [1, 1, 0.5], [0, 1, 0]], "coplanar"),
                          ([[0, 0, 0], [1, 0, 0], [0.2, 0.2, 0], [0, 1, 0]], "non-convex")):
        ply = tmp_path / f"{reason}.ply"
        head = ["ply", "format ascii 1.0", "element vertex 4"] + [f"property float {k}" for k in ("x", "y", "z", "nx", "ny", "nz")]
        head += ["element face 1", "property list uchar uint vertex_indices", "end_header"]
        rows = [" ".join(str(float(x)) for x in v) + " 0 0 1" for v in verts] + ["4 0 1 2 3"]
        ply.write_text("\n".join(head + rows) + "\n")
        scene = mesh_scene(tmp_path, str(ply))
        r = run("-s", scene, "--dump-scene", str(tmp_path / "o.bin"))
        assert r.returncode == 101 and reason in r.stderr
        with pytest.raises(config.ConfigError):
            config.load_scene(scene)


def test_cli_argument_errors(tmp_path):
    scene = os.path.join(ROOT, "configs", "smallpt", "scene.toml")
    render = os.path.join(ROOT, "configs", "smallpt", "render_high_quality.toml")
    assert run().returncode == 2
    r = run("-s", scene, "-r", render)
    assert r.returncode == 2 and "--output" in r.stderr
    assert run("-s", scene, "-r", render, "-o", "x.png", "--bogus").returncode == 2
    assert run("--help").returncode == 0 and "--scene-config" in run("--help").stdout
    assert run("-s", scene, "-r", render, "-o", "x.png", "--checkpoint-every", "4").returncode == 2  # needs --checkpoint
    assert run("-s", scene, "-r", render, "-o", "x.png", "--checkpoint", "c.bin", "--checkpoint-every", "0").returncode == 2
    # a well-formed call reaches the backend; without a GPU it must fail there, loudly, not fall back
    r = run("--scene-config=" + scene, "--render-config", render, "--output", str(tmp_path / "x.png"))
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 101 and "no HIP device" in r.stderr and not (tmp_path / "x.png").exists()


@pytest.mark.parametrize("text,msg", [
    ('[tracer]\ntype = "mcpt"\nmin_depth = 2\nmax_depth = 6\n', "missing field `image`"),
    ('[tracer]\ntype = "bdpt"\nmin_depth = 2\nmax_depth = 6\n[image]\nwidth=4\nheight=4\n[sampler]\ntype="Random"\n'
     'samples_per_pixel=1\n[post_processing]\ngamma_correction=true\n[performance]\n', "mcpt"),
    ('[tracer]\ntype = "mcpt"\nmin_depth = 2\nmax_depth = 6\n[image]\nwidth=4\nheight=4\n[sampler]\ntype="Stratified"\n'
     'samples_per_pixel=6\nx_strata=2\ny_strata=2\n[post_processing]\ngamma_correction=true\n[performance]\n',
     "x_strata * y_strata != samples_per_pixel"),
    ('[tracer]\ntype = "mcpt"\nmin_depth = 2.5\nmax_depth = 6\n[image]\nwidth=4\nheight=4\n[sampler]\ntype="Random"\n'
     'samples_per_pixel=1\n[post_processing]\ngamma_correction=true\n[performance]\n', "expected an integer"),
    ('[tracer\ntype = "mcpt"\n', "TOML parse error at line 1"),
    ('[tracer]\ntype = "mcpt"\ntype = "x"\n', "duplicate key"),
])
def test_render_config_errors(tmp_path, text, msg):
    rc = tmp_path / "r.toml"
    rc.write_text(text)
    r = run("-s", os.path.join(ROOT, "configs", "smallpt", "scene.toml"), "-r", str(rc), "-o", str(tmp_path / "x.png"))
    assert r.returncode == 101 and msg in r.stderr, r.stderr


def test_image_writer(tmp_path):
    from PIL import Image
    expect = np.zeros((21, 37, 3), dtype=np.uint8)
    for y in range(21):
        for x in range(37):
            for c in range(3):
                expect[y, x, c] = (x * 7 + y * 13 + c * 101) & 0xFF
    for ext in ("png", "ppm"):
        out = tmp_path / f"g.{ext}"
        assert run("--self-test-image", str(out)).returncode == 0
        assert np.array_equal(np.asarray(Image.open(out).convert("RGB")), expect)
    r = run("--self-test-image", str(tmp_path / "g.bmp"))
    assert r.returncode == 101 and "unsupported image format" in r.stderr


def test_host_and_oracle_under_sanitizers(tmp_path):
    """The whole of this file again on a build of the host with AddressSanitizer + UndefinedBehaviorSanitizer (every
    parser, loader and writer path above, malformed inputs included), and the oracle's renderer on a driver built the
    same way (oracle/san_check.c)."""
    if os.environ.get("RRT_HOST_BINARY"):
        pytest.skip("already running on another build of the host")
    import sys
    host = os.path.join(ROOT, "rustraytracer_amd", "host")
    san = os.path.join(ROOT, "tests", "c_abi", "rrt_san")
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O1", "-g", "-std=c++17", "-Wall", "-Wextra",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "rrt_main.cpp", "-o", san, "-pthread", "-L" + os.path.join(ROOT, "rustraytracer_amd", "csrc"), "-lrrt_hip",
                           "-lz", "-Wl,-rpath,$ORIGIN/../../rustraytracer_amd/csrc"], cwd=host)
    env = dict(os.environ, RRT_HOST_BINARY=san)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout and "skipped" in r.stdout
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san_check"])
    r = subprocess.run([os.path.join(ROOT, "oracle", "san_check")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "san_check ok" in r.stdout, r.stdout + r.stderr
