"""TOML schema, transforms and render-config handling of the host layer."""
import os

import numpy as np
import pytest

from conftest import ROOT
from rustraytracer_amd import _abi


def test_shipped_render_configs(cfg):
    expect = {
        "cornell_box/render_high_quality.toml": (500, 500, 1, 16384, 128, 128),
        "cornell_box/render_medium_quality.toml": (500, 500, 1, 1024, 32, 32),
        "cornell_box/render_low_quality.toml": (500, 500, 1, 64, 8, 8),
        "cornell_box/render_debug.toml": (250, 250, 0, 64, 0, 0),
        "smallpt/render_high_quality.toml": (1024, 768, 1, 16384, 128, 128),
        "smallpt/render_debug.toml": (320, 240, 1, 64, 8, 8),
    }
    for rel, (w, h, st, spp, xs, ys) in expect.items():
        rc = cfg.load_render_config(os.path.join(ROOT, "configs", rel))
        assert (rc.width, rc.height, rc.sampler_type, rc.spp, rc.x_strata, rc.y_strata) == (w, h, st, spp, xs, ys)
        assert (rc.min_depth, rc.max_depth) == (2, 6)
        assert rc.tone_mapping == 1 and rc.gamma_correction and rc.white_balance is None
        assert rc.parallelism == 0
        d = rc.render_desc()
        assert d.sample_begin == 0 and d.sample_end == (spp + 1 if st == 0 else spp)


def test_render_config_errors(cfg):
    base = {"tracer": {"type": "mcpt", "min_depth": 2, "max_depth": 6}, "image": {"width": 4, "height": 4},
            "sampler": {"type": "Stratified", "samples_per_pixel": 16, "x_strata": 4, "y_strata": 4},
            "post_processing": {"gamma_correction": False}, "performance": {}}
    rc = cfg.RenderConfig(base)
    assert rc.tone_mapping == 0 and rc.parallelism is None
    bad = dict(base, sampler={"type": "Stratified", "samples_per_pixel": 16, "x_strata": 4, "y_strata": 5})
    with pytest.raises(cfg.ConfigError, match="x_strata \\* y_strata != samples_per_pixel"):
        cfg.RenderConfig(bad)
    with pytest.raises(cfg.ConfigError):
        cfg.RenderConfig(dict(base, tracer={"type": "bdpt", "min_depth": 0, "max_depth": 1}))
    with pytest.raises(cfg.ConfigError):
        cfg.RenderConfig({k: v for k, v in base.items() if k != "performance"})
    unknown_tm = dict(base, post_processing={"tone_mapping": "aces", "gamma_correction": True,
                                             "white_balance": {"x": 1.0, "y": 0.9, "z": 0.8}})
    rc = cfg.RenderConfig(unknown_tm)
    assert rc.tone_mapping == 0 and np.allclose(rc.white_balance, [1.0, 0.9, 0.8])  # unknown names = identity


def test_matrix_is_column_major(cfg):
    # m41..m43 is the translation (cgmath Matrix4::new takes columns), src/math.rs:56-63
    t = {f"m{c}{r}": (1.0 if c == r else 0.0) for c in range(1, 5) for r in range(1, 5)}
    t.update(m41=1.0, m42=2.0, m43=3.0, m12=5.0)
    m = cfg.matrix_from_config(t)
    assert np.allclose(cfg.transform_point3(m, [0, 0, 0]), [1, 2, 3])
    assert np.allclose(cfg.transform_point3(m, [1, 0, 0]), [2, 7, 3])  # column 1 = (1, 5, 0)
    assert np.allclose(cfg.transform_vec3(m, [1, 0, 0]), [1, 5, 0])    # vectors ignore the translation


def test_shape_transforms(cfg):
    t = {f"m{c}{r}": (2.0 if c == r and c < 4 else 0.0) for c in range(1, 5) for r in range(1, 5)}
    t.update(m44=1.0, m41=1.0)
    scene = {"camera": {"type": "Perspective", "look_from": {"x": 0, "y": 0, "z": 0},
                        "look_at": {"x": 0, "y": 0, "z": -1}, "vup": {"x": 0, "y": 1, "z": 0}, "vfov": 90, "aspect": 1},
             "objects": [
                 {"shape": {"type": "Sphere", "center": {"x": 1, "y": 0, "z": 0}, "radius": 0.5, "transform": t},
                  "material": {"type": "IdealReflector"}},
                 {"shape": {"type": "Plane", "point": {"x": 0, "y": 1, "z": 0}, "normal": {"x": 0, "y": 3, "z": 0},
                            "transform": t}, "material": {"type": "Lambertian", "albedo": {"x": 1, "y": 1, "z": 1}}},
                 {"shape": {"type": "Triangle", "vertices": [{"x": 0, "y": 0, "z": 0}, {"x": 1, "y": 0, "z": 0},
                                                             {"x": 0, "y": 1, "z": 0}], "transform": t},
                  "material": {"type": "PhongSpecular", "specular": {"x": 1, "y": 1, "z": 1}, "shininess": 5}},
                 {"shape": {"type": "Quadrilateral", "vertices": [{"x": 0, "y": 0, "z": 0}, {"x": 1, "y": 0, "z": 0},
                                                                  {"x": 1, "y": 1, "z": 0}, {"x": 0, "y": 1, "z": 0}]},
                  "material": {"type": "IdealDielectric", "ior": 1.3}},
             ]}
    flat = cfg.scene_from_dict(scene)
    v = flat.prims["v"]
    assert np.allclose(v[0, :4], [3, 0, 0, 0.5])      # only the centre moves, the radius is not scaled
    assert np.allclose(v[1, :6], [1, 2, 0, 0, 1, 0])  # normal re-normalised after the transform
    assert np.allclose(v[2, :9], [1, 0, 0, 3, 0, 0, 1, 2, 0])
    assert np.allclose(v[3, :12], [0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 1, 0])
    assert list(flat.materials["type"]) == [3, 1, 2, 4] and flat.materials["param"][2] == 5 and flat.materials["param"][3] == 1.3


def test_scene_errors(cfg):
    cam = {"type": "Perspective", "look_from": {"x": 0, "y": 0, "z": 0}, "look_at": {"x": 0, "y": 0, "z": -1},
           "vup": {"x": 0, "y": 1, "z": 0}, "vfov": 90, "aspect": 1}
    with pytest.raises(cfg.ConfigError):
        cfg.scene_from_dict({"camera": dict(cam, type="Ortho"), "objects": []})
    with pytest.raises(cfg.ConfigError):
        cfg.scene_from_dict({"camera": cam, "objects": [
            {"shape": {"type": "Torus"}, "material": {"type": "IdealReflector"}}]})
    with pytest.raises(cfg.ConfigError):
        cfg.scene_from_dict({"camera": cam, "objects": [
            {"shape": {"type": "Sphere", "center": {"x": 0, "y": 0, "z": 0}, "radius": 1}, "material": {"type": "Wax"}}]})
    with pytest.raises(cfg.ConfigError, match="cannot open mesh"):
        cfg.scene_from_dict({"camera": cam, "objects": [
            {"shape": {"type": "Mesh", "file": "does/not/exist.ply"}, "material": {"type": "IdealReflector"}}]})


def test_struct_layouts_match_header():
    import ctypes as C

    assert C.sizeof(_abi.Camera) == 96
    assert C.sizeof(_abi.SceneDesc) == 96 + 8 + 16
    assert C.sizeof(_abi.RenderDesc) == 72
    assert C.sizeof(_abi.PostDesc) == 40
    assert C.sizeof(_abi.Stats) == 8 * 4 + 40 + 8 + 16 + 16 + 8 + 16
    assert _abi.PRIM_DTYPE.itemsize == 112 and _abi.MATERIAL_DTYPE.itemsize == 40


def test_malformed_files_fail_one_way(tmp_path, cfg, monkeypatch):
    """Whatever a loader trips over in a damaged file, it raises ConfigError (every such input is one panic in the
    reference): bytes that are not UTF-8, a mesh cut off in the middle, a property line cut short, an element count no
    file could hold."""
    import shutil
    shutil.copytree(os.path.join(ROOT, "configs", "cornell_box"), tmp_path / "configs" / "cornell_box")
    monkeypatch.chdir(tmp_path)  # (mesh paths resolve against the working directory first, as in the reference)
    scene = tmp_path / "configs" / "cornell_box" / "scene.toml"
    ply = tmp_path / "configs" / "cornell_box" / "cornell_box_0000_m000.ply"
    good_scene, good_ply = scene.read_bytes(), ply.read_bytes()
    base = (str(tmp_path), str(tmp_path / "configs" / "cornell_box"))
    assert cfg.load_scene(str(scene), base_dirs=base).n_prims == 18
    for damaged in (good_scene[:300] + b"\xe8\xff" + good_scene[300:], good_scene[: len(good_scene) // 2], good_scene.replace(b"Mesh", b"Mash", 1)):
        scene.write_bytes(damaged)
        with pytest.raises(cfg.ConfigError):
            cfg.load_scene(str(scene), base_dirs=base)
    scene.write_bytes(good_scene)
    head_end = good_ply.index(b"end_header")
    for damaged in (good_ply[: head_end + 40], good_ply.replace(b"property float x", b"property float", 1),
                    good_ply.replace(b"element vertex ", b"element vertex 99999999", 1), good_ply.replace(b"float", b"\x02loat", 1)):
        ply.write_bytes(damaged)
        with pytest.raises(cfg.ConfigError):
            cfg.load_scene(str(scene), base_dirs=base)
    render = tmp_path / "r.toml"
    render.write_bytes(b"[tracer]\ntype = \"mcpt\"\nmin_depth = \xff\n")
    with pytest.raises(cfg.ConfigError):
        cfg.load_render_config(str(render))
