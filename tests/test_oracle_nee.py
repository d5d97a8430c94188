"""Next-event estimation (RRT_RENDER_NEE) in the CPU oracle.  The reference has no such estimator (its connect()
panics for s > 0, src/tracers/utils.rs:122-124), so there is nothing of the reference's to pin it on: the checks
are that it is the SAME integral as the reference's estimator (equal means within Monte-Carlo error, exact on
analytic scenes in expectation) with less noise for a small light."""
import numpy as np
import pytest

import scenes_nee as N
import scenes_util as S
from rustraytracer_amd import _abi


def _mean(rgb, spp):
    return rgb.reshape(-1, 3).mean(axis=0) / spp


@pytest.mark.parametrize("light", ["pgram", "triangle", "sphere"])
def test_same_mean_as_the_reference_estimator(O, cfg, light):
    flat = N.lit_room(light)
    rc = cfg.stratified_config(24, 18, 32, 32, min_depth=2, max_depth=6, seed=5)
    plain, st0 = O.render_accumulate(flat, rc.render_desc())
    nee, st1 = O.render_accumulate(flat, rc.render_desc(flags=_abi.RENDER_NEE))
    assert st0["samples"] == st1["samples"] and st1["nonfinite"] == 0
    assert st1["rays"] > st0["rays"]  # the shadow rays are counted
    m0, m1 = _mean(plain, rc.spp), _mean(nee, rc.spp)
    # 440 k samples: over eight seeds the plain estimator's mean scatters by 1.6 % here and the other one's by 0.8 %
    # (what is left is the caustic of the glass ball and the light seen directly), their averages agree to 0.1 %.
    assert np.allclose(m1, m0, rtol=0.06), (m0, m1)


def test_less_noise_with_a_small_light(O, cfg):
    flat = N.lit_room("pgram")
    ref, _ = O.render_accumulate(flat, cfg.stratified_config(16, 12, 48, 48, seed=1).render_desc(flags=_abi.RENDER_NEE))
    ref = ref / (48 * 48)
    rc = cfg.stratified_config(16, 12, 6, 6, seed=2)
    plain, _ = O.render_accumulate(flat, rc.render_desc())
    nee, _ = O.render_accumulate(flat, rc.render_desc(flags=_abi.RENDER_NEE))
    e0 = np.sqrt(np.mean((plain / rc.spp - ref) ** 2))
    e1 = np.sqrt(np.mean((nee / rc.spp - ref) ** 2))
    assert e1 < 0.6 * e0, (e0, e1)


def test_enclosing_sphere_light_gives_albedo_times_L(O, cfg):
    """Lambertian floor inside an emitting sphere: the plain estimator gives albedo x L per sample exactly; sampling
    the sphere's surface instead must give it in the mean (the light is seen from inside: emitters radiate from
    both sides)."""
    albedo = (0.8, 0.5, 0.25)
    flat = S.lambert_floor(albedo)
    rc = cfg.stratified_config(12, 12, 24, 24, seed=3)
    rgb, st = O.render_accumulate(flat, rc.render_desc(flags=_abi.RENDER_NEE))
    on_floor = (O.trace_primary(flat, *np.meshgrid((np.arange(12) + 0.5) / 12, 1 - (np.arange(12) + 0.5) / 12))[0] == 1)
    px = (rgb / rc.spp).reshape(-1, 3)[on_floor.reshape(-1)]
    assert len(px) > 40
    assert np.allclose(px.mean(axis=0), np.array(albedo) * np.array(S.L), rtol=0.02)


def test_every_material_same_mean(O, cfg):
    flat = S.all_prims_scene()
    rc = cfg.stratified_config(24, 18, 12, 12, min_depth=1, max_depth=8, seed=9)
    plain, _ = O.render_accumulate(flat, rc.render_desc())
    nee, _ = O.render_accumulate(flat, rc.render_desc(flags=_abi.RENDER_NEE))
    assert np.allclose(_mean(nee, rc.spp), _mean(plain, rc.spp), rtol=0.03)


def test_emissive_plane_is_refused(O, cfg):
    with pytest.raises(ValueError):
        O.render_accumulate(N.emissive_plane_room(), cfg.stratified_config(4, 4, 2, 2).render_desc(flags=_abi.RENDER_NEE))


def test_flag_off_is_the_reference_estimator(O, cfg, cornell):
    rc = cfg.stratified_config(16, 16, 4, 4, seed=11)
    a, _ = O.render_accumulate(cornell, rc.render_desc(), device_numbers=True)
    b, _ = O.render_accumulate(cornell, rc.render_desc(flags=0), device_numbers=True)
    assert np.array_equal(a, b)


def test_bent_quad_light_from_below(O, cfg):
    """An emitting quad that is not planar is sampled as its two triangles.  The reference's quad test
    (src/shapes/quadrilateral.rs:56-117) is those two triangles for rays that arrive steeply, but from grazing
    directions it accepts rays far outside them (the corner test at v2 is an unbounded wedge evaluated on another
    plane: seen from (-3.9, 4.8, -5) this quad subtends 4.5 times the solid angle of its triangles), so the two
    estimators can only agree where the light is seen steeply: here, on the floor under it."""
    flat = N.lit_room("bent")
    rc = cfg.stratified_config(24, 18, 32, 32, min_depth=0, max_depth=2, seed=5)  # direct light only
    plain, _ = O.render_accumulate(flat, rc.render_desc())
    nee, _ = O.render_accumulate(flat, rc.render_desc(flags=_abi.RENDER_NEE))
    u, v = np.meshgrid((np.arange(24) + 0.5) / 24, 1 - (np.arange(18) + 0.5) / 18)
    oid, pid, _ = O.trace_primary(flat, u, v)
    floor = ((oid == 0) & (pid == 0)).reshape(18, 24)
    assert floor.sum() > 50
    m0, m1 = plain[floor].mean(axis=0), nee[floor].mean(axis=0)
    assert np.allclose(m1, m0, rtol=0.04), (m0, m1)
