"""The f64 device renderer against the oracle PATH BY PATH.

The oracle can draw its random numbers from a restatement of the device's counter-based generator
(oracle.c, "device numbers"); the f64 kernels build the reference's orthonormal basis
(src/math.rs:162-172).  Both then trace the same paths for the same seed, and every pixel sum must
agree to rounding: the whole loop of generate_camera_vertices (src/tracers/utils.rs:16-91) -- closest
hit, emission, Russian roulette, the sampler's dimension contract, scatter, throughput -- is compared
deterministically instead of statistically.

What may differ: the device accumulates in f64 per work item and stores f32 (one rounding per item);
a ray within an ulp of an edge, a Russian-roulette or Fresnel threshold can go the other way (the
device's intersection tests are algebraically equal to the reference's, not bitwise), which changes
that one pixel visibly -- the tests allow a handful of such pixels per image."""
import numpy as np
import pytest

import scenes_util as S

pytestmark = pytest.mark.gpu


def _compare(api, be, O, cfg, flat, rc, max_bad, accel=None, pipeline=None):
    sc = api.Scene(flat)
    dev, st = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F64, accel=api.ACCEL_AUTO if accel is None else accel,
                                                       pipeline=api.PIPELINE_AUTO if pipeline is None else pipeline))
    if pipeline is not None:
        assert st["pipeline_used"] == pipeline
    ref, rst = O.render_accumulate(flat, rc.render_desc(), device_numbers=True)
    assert st["samples"] == rst["samples"] and st["nonfinite"] == 0 and rst["nonfinite"] == 0
    err = np.abs(dev.astype(np.float64) - ref)
    # f32 storage of (a few) f64 item sums; the wavefront pipeline adds every sample's radiance to the f32 sum by itself
    tol = (4e-7 if pipeline != api.PIPELINE_WAVEFRONT else 3e-7 * np.sqrt(rc.spp)) * np.abs(ref) + 1e-9
    bad = int(np.any(err > tol, axis=-1).sum())
    assert bad <= max_bad, (bad, float(err.max()))
    # the same paths: the same number of closest-hit queries (the oracle also counts queries of paths whose
    # throughput has dropped to exactly zero, which the device drops) and of scatter events per material
    assert abs(st["rays"] - rst["rays_live"]) <= 8 * max(max_bad, 1)
    return bad


@pytest.mark.parametrize("name", ["cornell_box", "smallpt"])
def test_shipped_scenes_same_paths(gpu, O, cfg, scenes, name):
    api, be = gpu
    for rc in (cfg.stratified_config(96, 72, 8, 8, seed=17), cfg.stratified_config(40, 40, 5, 3, seed=18),
               cfg.random_config(64, 48, 24, seed=19)):
        bad = _compare(api, be, O, cfg, scenes[name], rc, max_bad=3)
        print(name, rc.spp, "pixels off:", bad)


def test_every_primitive_and_material_same_paths(gpu, O, cfg):
    """Sphere, plane, triangle, quad and mesh faces; dielectric, Lambertian, Phong, mirror, emitter."""
    api, be = gpu
    flat = S.all_prims_scene()
    for rc in (cfg.stratified_config(80, 60, 6, 6, min_depth=1, max_depth=8, seed=23),
               cfg.random_config(48, 48, 16, min_depth=0, max_depth=5, seed=24)):
        _compare(api, be, O, cfg, flat, rc, max_bad=4)


def test_wavefront_pipeline_in_f64_same_paths(gpu, O, cfg, scenes, cornell):
    """The queue pipeline in the reference's arithmetic type: the one-kernel-per-trip form of scan scenes and the
    closest-hit + shade pair of tree scenes trace the oracle's paths, like the fused kernel."""
    api, be = gpu
    wf = api.PIPELINE_WAVEFRONT
    for name in ("cornell_box", "smallpt"):
        bad = _compare(api, be, O, cfg, scenes[name], cfg.stratified_config(96, 72, 6, 6, seed=41), max_bad=3, accel=api.ACCEL_LINEAR, pipeline=wf)
        print(name, "scan, pixels off:", bad)
    _compare(api, be, O, cfg, cornell, cfg.random_config(64, 48, 24, seed=42), max_bad=3, accel=api.ACCEL_LINEAR, pipeline=wf)
    for tree in (api.ACCEL_LBVH, api.ACCEL_SAH_GPU):
        _compare(api, be, O, cfg, cornell, cfg.stratified_config(64, 64, 4, 4, seed=43), max_bad=3, accel=tree, pipeline=wf)
        _compare(api, be, O, cfg, S.all_prims_scene(), cfg.stratified_config(64, 48, 4, 4, min_depth=1, max_depth=8, seed=44), max_bad=4,
                 accel=tree, pipeline=wf)
    flat = _two_mesh_scene(5000)
    bad = _compare(api, be, O, cfg, flat, cfg.stratified_config(64, 48, 3, 3, seed=45), max_bad=4, accel=api.ACCEL_SAH_GPU, pipeline=wf)
    print("10 000 triangles, pixels off:", bad)


def test_same_paths_through_the_tree(gpu, O, cfg, cornell):
    """Closest hit through either tree instead of the scan: the same primitive and t for every ray, so the same image."""
    api, be = gpu
    for tree in (api.ACCEL_LBVH, api.ACCEL_SAH, api.ACCEL_SAH_GPU):
        _compare(api, be, O, cfg, cornell, cfg.stratified_config(64, 64, 4, 4, seed=31), max_bad=3, accel=tree)
        _compare(api, be, O, cfg, S.all_prims_scene(), cfg.stratified_config(64, 48, 4, 4, max_depth=8, seed=32), max_bad=4,
                 accel=tree)


def test_sample_range_shards_same_paths(gpu, O, cfg, cornell):
    """A shard of the sample index range traces exactly those samples of the oracle's render."""
    api, be = gpu
    sc = api.Scene(cornell)
    rc = cfg.stratified_config(48, 48, 4, 4, seed=29)
    ref, _ = O.render_accumulate(cornell, rc.render_desc(), device_numbers=True)
    acc = None
    for lo, hi in ((0, 5), (5, 16)):
        acc, _ = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F64, sample_begin=lo, sample_end=hi), acc)
    err = np.abs(acc.astype(np.float64) - ref)
    assert int(np.any(err > 6e-7 * np.abs(ref) + 1e-9, axis=-1).sum()) <= 2


def _follow(api, be, O, flat, rc, pipeline, label):
    """Fraction of pixels in which the f32 product kernels leave the oracle's paths, and the error elsewhere."""
    sc = api.Scene(flat)
    dev, st = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F32, pipeline=pipeline))
    ref, rst = O.render_accumulate(flat, rc.render_desc(), device_numbers="f32")
    assert st["samples"] == rst["samples"] and st["nonfinite"] == 0
    err = np.abs(dev.astype(np.float64) - ref)
    # f32 arithmetic along a path: throughput products, 1 / q of Russian roulette (v_rcp_f32), f32 accumulation
    off = np.any(err > 2e-5 * np.abs(ref) + 1e-6, axis=-1)
    rel_rays = abs(st["rays"] - rst["rays_live"]) / rst["rays_live"]
    print(label, "pixels off: %d of %d (%.2f %%), closest-hit queries differ by %.2e" % (off.sum(), off.size, 100 * off.mean(), rel_rays))
    return off.mean(), rel_rays


@pytest.mark.parametrize("name", ["cornell_box", "smallpt"])
def test_f32_kernels_follow_the_same_paths(gpu, O, cfg, scenes, name):
    """The product kernels (f32, fused and wavefront pipelines) against the oracle on THEIR numbers (24-bit uniforms,
    their lobe basis): a path leaves the oracle's only where f32 rounding flips a decision -- a ray within ~1e-6 of an
    edge or of the light's outline, a roulette or Fresnel draw within ~1e-7 of its threshold -- and then the pixel is
    off by a whole sample; everywhere else the sums agree to f32 rounding.  Measured at 64 spp: no pixel of the Cornell
    box, 4 of 6912 of smallpt (sphere silhouettes).  Gate: 0.5 % of the pixels, closest-hit query counts within 1e-4."""
    api, be = gpu
    for rc in (cfg.stratified_config(96, 72, 8, 8, seed=41), cfg.random_config(64, 48, 31, min_depth=1, max_depth=9, seed=42)):
        for pipeline, label in ((api.PIPELINE_FUSED, "fused"), (api.PIPELINE_WAVEFRONT, "wavefront")):
            frac, rel_rays = _follow(api, be, O, scenes[name], rc, pipeline, "%s %s %d spp" % (name, label, rc.spp))
            assert frac <= 0.005 and rel_rays <= 1e-4


@pytest.mark.parametrize("case", ["cornell_full_image", "cornell_full_spp", "smallpt_full_image", "cornell_cube"])
def test_f32_kernels_at_benchmark_shapes(gpu, O, cfg, scenes, case):
    """The benchmark configurations with ONE of their two sizes reduced so that the oracle finishes in a second:
    the full 500x500 / 1024x768 images at 4 spp, and all 16384 samples per pixel (128 x 128 strata, what the
    headline number is quoted on) of a 16x16 image."""
    import os
    from conftest import ROOT
    api, be = gpu
    flat, rc = {
        "cornell_full_image": (scenes["cornell_box"], cfg.stratified_config(500, 500, 2, 2, seed=51)),
        "cornell_full_spp": (scenes["cornell_box"], cfg.stratified_config(16, 16, 128, 128, seed=52)),
        "smallpt_full_image": (scenes["smallpt"], cfg.stratified_config(1024, 768, 2, 2, seed=53)),
        "cornell_cube": (cfg.load_scene(os.path.join(ROOT, "configs", "cornell_cube", "scene.toml")),
                         cfg.stratified_config(256, 256, 8, 8, seed=54)),
    }[case]
    frac, rel_rays = _follow(api, be, O, flat, rc, api.PIPELINE_AUTO, case)
    n_pixels, n_samples = rc.width * rc.height, rc.width * rc.height * rc.spp
    off_pixels = frac * n_pixels
    assert rel_rays <= 1e-4
    # Two gates, each where it means something.  A decision flipped by f32 rounding costs one SAMPLE and marks the pixel
    # that holds it.  (1) Per sample: at most 15 flips per million samples (measured: 0.5 per million in the Cornell
    # box at 16384 spp, 11 per million on smallpt's sphere silhouettes at 4 spp).  (2) Per pixel: at most 0.5 % of
    # the pixels -- asked wherever a pixel holds few samples; a pixel that sums 16384 samples is marked by a single
    # flip among them, so there the count of marked pixels IS the count of flips and gate (1) is the one that applies.
    assert off_pixels <= 15e-6 * n_samples + 0.5, (case, off_pixels, n_samples)
    if rc.spp <= 64:
        assert frac <= 0.005, (case, frac)
    else:
        assert off_pixels <= 2e-6 * n_samples, (case, off_pixels)


@pytest.mark.parametrize("name", ["cornell_box", "smallpt"])
def test_committed_renders(gpu, cfg, scenes, name):
    """The device against tests/golden/paths.npz (made by tools/make_golden.py from the oracle): no oracle at run time."""
    import os
    from conftest import GOLDEN
    api, be = gpu
    g = np.load(os.path.join(GOLDEN, "paths.npz"))
    sc = api.Scene(scenes[name])
    rc = cfg.stratified_config(32, 32, 4, 4, seed=61)
    for mode, prec, rel, max_bad in (("f64", api.PRECISION_F64, 4e-7, 1), ("f32", api.PRECISION_F32, 2e-5, 5)):
        dev, st = be.render_accumulate(sc, rc.render_desc(precision=prec))
        ref = g[f"{name}_{mode}"]
        bad = int(np.any(np.abs(dev - ref) > rel * np.abs(ref) + 1e-6, axis=-1).sum())
        assert bad <= max_bad, (mode, bad)
        assert abs(st["rays"] - int(g[f"{name}_{mode}_rays"])) <= 8 * max_bad


def _two_mesh_scene(n_tri_each):
    """Two displaced spheres of triangles -- glass (ior 1.5) and Phong, the materials of the million-triangle
    benchmark scene -- over a Lambertian plane inside an emissive sphere."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_mesh
    from rustraytracer_amd import _abi, config
    parts = []
    for k, (centre, seed) in enumerate((((-1.1, 0.0, -4.0), 3), ((1.1, 0.1, -4.5), 4))):
        v, _, f = make_mesh.displaced_sphere(n_tri_each, radius=1.0, bump=0.15, seed=seed)
        tri = v[f].astype(np.float64) + np.array(centre)
        p = np.zeros(len(f), dtype=_abi.PRIM_DTYPE)
        p["type"], p["object_id"], p["prim_id"] = _abi.PRIM_TRIANGLE, k, np.arange(len(f))
        p["v"][:, :9] = tri.reshape(len(f), 9)
        parts.append(p)
    rest = np.zeros(2, dtype=_abi.PRIM_DTYPE)
    rest["type"][0], rest["object_id"][0] = _abi.PRIM_SPHERE, 2
    rest["v"][0, :4] = [0, 0, 0, 40.0]
    rest["type"][1], rest["object_id"][1] = _abi.PRIM_PLANE, 3
    rest["v"][1, :6] = [0, -1.3, 0, 0, 1, 0]
    m = np.zeros(4, dtype=_abi.MATERIAL_DTYPE)
    for i, (typ, rgb, param) in enumerate(((_abi.MAT_DIELECTRIC, (0, 0, 0), 1.5), (_abi.MAT_PHONG, (0.8, 0.7, 0.3), 25.0),
                                           (_abi.MAT_EMISSIVE, (1.0, 0.9, 0.8), 0.0), (_abi.MAT_LAMBERTIAN, (0.6, 0.6, 0.6), 0.0))):
        m["type"][i], m["rgb"][i], m["param"][i] = typ, rgb, param
    cam = config.perspective_camera((0, 0.3, 0), (0, 0, -4), (0, 1, 0), 45.0, 1.0)
    return config.FlatScene(cam, np.concatenate(parts + [rest]), m)


def test_mesh_scene_through_tree_and_wavefront_same_paths(gpu, O, cfg):
    """20 000 triangles through the tree AUTO builds, with the fused kernel (f64) and with the wavefront pipeline
    (f32: persistent tree-walk kernel, ray queues, compaction; what AUTO takes for larger scenes); the oracle scans
    every triangle for every ray.  Same paths, so the same pixels."""
    api, be = gpu
    flat = _two_mesh_scene(10000)
    assert flat.n_prims > 15000
    rc = cfg.stratified_config(64, 64, 2, 2, max_depth=8, seed=71)
    bad = _compare(api, be, O, cfg, flat, rc, max_bad=4)                      # f64: fused kernel + tree
    frac, rel_rays = _follow(api, be, O, flat, rc, api.PIPELINE_WAVEFRONT, "two meshes")  # f32: wavefront + tree
    print("f64 pixels off:", bad)
    assert frac <= 0.01 and rel_rays <= 3e-4


def _nasty_scene():
    """Coincident surfaces with different materials (the scan order decides which one a ray sees), a zero-area
    triangle, a quad a millimetre above a plane, an albedo above one, Phong exponents 0 and 1000, glass with index 1."""
    from rustraytracer_amd import _abi
    q = [[-2, -1, -6], [0, -1, -6], [0, 1, -6], [-2, 1, -6]]
    prims = [
        (3, 0, 0, q), (3, 1, 0, q),                                        # the same quad twice: the later object wins the tie
        (2, 2, 0, [[0.5, -1, -5], [2.5, -1, -5], [1.5, 1, -5]]),
        (2, 3, 0, [[0.5, -1, -5], [2.5, -1, -5], [1.5, 1, -5]]),           # the same triangle twice: the earlier one wins
        (2, 2, 1, [[0, 0, -3], [0, 0, -3], [0, 0, -3]]),                   # zero area
        (1, 4, 0, [0, -1, 0, 0, 1, 0]),                                     # floor plane ...
        (3, 5, 0, [[-3, -0.999, -2], [3, -0.999, -2], [3, -0.999, -8], [-3, -0.999, -8]]),  # ... and a quad a millimetre above it
        # (a quad IN the plane would be a tie between two different formulas for t, which rounding decides)
        (0, 6, 0, [1.8, 0.2, -3.5, 0.7]),                                   # glass with ior 1
        (0, 7, 0, [-1.2, -0.3, -3.0, 0.6]),                                 # Phong, exponent 1000
        (0, 8, 0, [0, 0, 0, 25.0]),                                         # light all around (the camera is inside)
    ]
    mats = [(_abi.MAT_LAMBERTIAN, (0.2, 0.9, 0.2), 0.0), (_abi.MAT_LAMBERTIAN, (1.3, 0.4, 0.4), 0.0),
            (_abi.MAT_REFLECTOR, (0, 0, 0), 0.0), (_abi.MAT_LAMBERTIAN, (0.1, 0.1, 0.9), 0.0),
            (_abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0), (_abi.MAT_PHONG, (0.9, 0.9, 0.9), 0.0),
            (_abi.MAT_DIELECTRIC, (0, 0, 0), 1.0), (_abi.MAT_PHONG, (0.8, 0.8, 0.8), 1000.0),
            (_abi.MAT_EMISSIVE, (0.9, 1.0, 1.1), 0.0)]
    return S.make_scene(prims, mats, look_from=(0, 0.4, 1.5), look_at=(0, 0, -5), vfov=50.0, aspect=4 / 3)


def test_ties_and_degenerate_inputs_same_paths(gpu, O, cfg):
    api, be = gpu
    flat = _nasty_scene()
    rc = cfg.stratified_config(96, 72, 4, 4, min_depth=1, max_depth=10, seed=81)
    for accel in (api.ACCEL_LINEAR, api.ACCEL_LBVH, api.ACCEL_SAH, api.ACCEL_SAH_GPU):
        bad = _compare(api, be, O, cfg, flat, rc, max_bad=6, accel=accel)
        print("accel", accel, "f64 pixels off:", bad)
    frac, rel_rays = _follow(api, be, O, flat, rc, api.PIPELINE_FUSED, "nasty fused")
    assert frac <= 0.005 and rel_rays <= 1e-4
    frac, rel_rays = _follow(api, be, O, flat, rc, api.PIPELINE_WAVEFRONT, "nasty wavefront")
    assert frac <= 0.005 and rel_rays <= 1e-4


def test_million_triangle_scene_same_paths(gpu, O, cfg):
    """configs/mesh_1m (998 290 primitives): the oracle, whose scan needs ~10 ms per ray here, goes through its own
    bounding-volume hierarchy (tests/test_oracle_accel.py: the same hits as its scan) and draws the device's numbers;
    the device renders through the GPU-built LBVH.  f64 fused kernel: the same image pixel by pixel; f32 product path
    (wavefront pipeline, what the benchmark runs): the same paths up to f32 rounding flips."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_c5_assets
    make_c5_assets.main()
    api, be = gpu
    flat = cfg.load_scene(os.path.join(ROOT, "configs", "mesh_1m", "scene.toml"))
    assert flat.n_prims > 900000
    sc = api.Scene(flat)
    rc = cfg.stratified_config(192, 108, 2, 2, seed=71)
    with O.accelerated(flat):
        ref64, rst64 = O.render_accumulate(flat, rc.render_desc(), device_numbers=True)
        ref32, rst32 = O.render_accumulate(flat, rc.render_desc(), device_numbers="f32")
    dev, st = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F64, accel=api.ACCEL_SAH))
    err = np.abs(dev.astype(np.float64) - ref64)
    bad = int(np.any(err > 4e-7 * np.abs(ref64) + 1e-9, axis=-1).sum())
    print("mesh_1m f64: pixels off %d of %d, rays %d / %d" % (bad, rc.width * rc.height, st["rays"], rst64["rays_live"]))
    assert bad <= 6 and abs(st["rays"] - rst64["rays_live"]) <= 64
    dev, st = be.render_accumulate(sc, rc.render_desc())  # AUTO: tree + wavefront pipeline, f32
    off = np.any(np.abs(dev.astype(np.float64) - ref32) > 2e-5 * np.abs(ref32) + 1e-6, axis=-1)
    rel_rays = abs(st["rays"] - rst32["rays_live"]) / rst32["rays_live"]
    print("mesh_1m f32: pixels off %d of %d (%.2f %%), closest-hit queries differ by %.2e" % (off.sum(), off.size, 100 * off.mean(), rel_rays))
    assert off.mean() <= 0.01 and rel_rays <= 1e-3
