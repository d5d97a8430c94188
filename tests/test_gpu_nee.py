"""Next-event estimation (RRT_RENDER_NEE) on the device against its CPU statement in the oracle, PATH BY PATH: the
oracle draws the device's numbers, so both sample the same lights at the same points and trace the same shadow rays.
(The reference has no such estimator, src/tracers/utils.rs:122-124; tests/test_oracle_nee.py checks on the CPU that
it computes the same image as the reference's estimator.)"""
import numpy as np
import pytest

import scenes_nee as N
import scenes_util as S

pytestmark = pytest.mark.gpu


def _compare(api, be, O, flat, rc, precision, max_bad, accel=None, pipeline=None):
    from rustraytracer_amd import _abi
    sc = api.Scene(flat)
    f64 = precision == api.PRECISION_F64
    dev, st = be.render_accumulate(sc, rc.render_desc(precision=precision, flags=_abi.RENDER_NEE,
                                                       accel=api.ACCEL_AUTO if accel is None else accel,
                                                       pipeline=api.PIPELINE_FUSED if pipeline is None else pipeline))
    assert st["pipeline_used"] == (api.PIPELINE_FUSED if pipeline is None else pipeline)
    ref, rst = O.render_accumulate(flat, rc.render_desc(flags=_abi.RENDER_NEE), device_numbers=True if f64 else "f32")
    assert st["samples"] == rst["samples"] and st["nonfinite"] == 0 and rst["nonfinite"] == 0
    err = np.abs(dev.astype(np.float64) - ref)
    # (the wavefront pipeline adds every contribution to the f32 sum by itself, the fused kernel a work item's f64 sum)
    tol = ((3e-7 * np.sqrt(2 * rc.spp) if pipeline == api.PIPELINE_WAVEFRONT else 4e-7) if f64 else 5e-5) * np.abs(ref) + (1e-9 if f64 else 1e-5)
    bad = int(np.any(err > tol, axis=-1).sum())
    assert bad <= max_bad, (bad, float(err.max()))
    # the same camera, scattered AND shadow rays (the oracle also counts rays of paths whose throughput is zero)
    print("rays", st["rays"], rst["rays_live"], rst["rays"], "bounces", st["bounces"], rst["bounces"])
    assert abs(st["rays"] - rst["rays_live"]) <= (8 if f64 else 64) * max(max_bad, 1)
    return bad


@pytest.mark.parametrize("light", ["pgram", "triangle", "bent", "sphere"])
def test_same_paths_f64(gpu, O, cfg, light):
    api, be = gpu
    flat = N.lit_room(light)
    for rc in (cfg.stratified_config(64, 48, 6, 6, min_depth=2, max_depth=6, seed=41), cfg.random_config(48, 36, 20, seed=42)):
        print(light, "pixels off:", _compare(api, be, O, flat, rc, api.PRECISION_F64, max_bad=4))


@pytest.mark.parametrize("pipeline", ["fused", "wavefront"])
def test_same_paths_f32(gpu, O, cfg, cornell, pipeline):
    """The product kernels, both pipelines: the fused kernel traces the shadow ray in the trip of its vertex, the
    wavefront pipeline on the spot too for scan scenes and through a queue of shadow rays for tree scenes."""
    api, be = gpu
    pl = api.PIPELINE_FUSED if pipeline == "fused" else api.PIPELINE_WAVEFRONT
    rc = cfg.stratified_config(64, 48, 6, 6, seed=43)
    # f32: a shadow ray that grazes an edge, or a light pick within an ulp of a table entry, goes the other way
    for flat in (N.lit_room("pgram"), N.lit_room("sphere"), N.lit_room("bent"), cornell):
        for accel in (api.ACCEL_LINEAR, api.ACCEL_SAH_GPU):
            print(pipeline, accel, "pixels off:", _compare(api, be, O, flat, rc, api.PRECISION_F32, max_bad=24, accel=accel, pipeline=pl))


def test_wavefront_every_material_and_small_queues(gpu, O, cfg, monkeypatch):
    """Every material, a tree, and queues far smaller than the frame (many trips, shadow queue refilled every trip)."""
    api, be = gpu
    monkeypatch.setenv("RRT_DEBUG_WF_CAPACITY", "4096")
    rc = cfg.stratified_config(48, 36, 4, 4, min_depth=1, max_depth=8, seed=44)
    for accel in (api.ACCEL_LINEAR, api.ACCEL_LBVH):
        _compare(api, be, O, S.all_prims_scene(), rc, api.PRECISION_F32, max_bad=24, accel=accel, pipeline=api.PIPELINE_WAVEFRONT)


def test_every_material_and_the_tree(gpu, O, cfg, cornell):
    api, be = gpu
    rc = cfg.stratified_config(48, 36, 4, 4, min_depth=1, max_depth=8, seed=44)
    _compare(api, be, O, S.all_prims_scene(), rc, api.PRECISION_F64, max_bad=4)
    _compare(api, be, O, S.all_prims_scene(), rc, api.PRECISION_F64, max_bad=4, accel=api.ACCEL_LBVH)
    _compare(api, be, O, cornell, cfg.stratified_config(48, 48, 4, 4, seed=45), api.PRECISION_F64, max_bad=3, accel=api.ACCEL_SAH)
    # the queue pipeline in f64: shadow rays on the spot (scan) and through the shadow queue (tree)
    for accel in (api.ACCEL_LINEAR, api.ACCEL_SAH_GPU):
        _compare(api, be, O, S.all_prims_scene(), rc, api.PRECISION_F64, max_bad=4, accel=accel, pipeline=api.PIPELINE_WAVEFRONT)
        _compare(api, be, O, cornell, cfg.stratified_config(48, 48, 4, 4, seed=46), api.PRECISION_F64, max_bad=3, accel=accel, pipeline=api.PIPELINE_WAVEFRONT)


def test_converges_to_the_reference_estimator(gpu, cfg, cornell):
    """The product kernels, Cornell box: same mean image with and without the flag, and less noise with it."""
    from rustraytracer_amd import _abi
    api, be = gpu
    sc = api.Scene(cornell)
    big = cfg.stratified_config(64, 64, 64, 64, seed=46)
    ref, _ = be.render_accumulate(sc, big.render_desc())
    ref = ref / big.spp
    rc = cfg.stratified_config(64, 64, 8, 8, seed=47)
    plain, st0 = be.render_accumulate(sc, rc.render_desc())
    nee, st1 = be.render_accumulate(sc, rc.render_desc(flags=_abi.RENDER_NEE))
    assert st1["rays"] > st0["rays"] and st1["nonfinite"] == 0
    assert np.allclose(nee.mean(axis=(0, 1)) / rc.spp, ref.mean(axis=(0, 1)), rtol=0.02)
    e0 = np.sqrt(np.mean((plain / rc.spp - ref) ** 2))
    e1 = np.sqrt(np.mean((nee / rc.spp - ref) ** 2))
    print("RMSE at 64 spp: plain %.4f, next-event %.4f" % (e0, e1))
    assert e1 < 0.7 * e0


def test_refusals(gpu, cfg, cornell):
    from rustraytracer_amd import _abi
    api, be = gpu
    rc = cfg.stratified_config(16, 16, 2, 2)
    with pytest.raises(api.RrtError):  # an emitting plane has no area to sample
        be.render_accumulate(api.Scene(N.emissive_plane_room()), rc.render_desc(flags=_abi.RENDER_NEE))
    with pytest.raises(api.RrtError):
        be.render_accumulate(api.Scene(cornell), rc.render_desc(flags=0x80))
