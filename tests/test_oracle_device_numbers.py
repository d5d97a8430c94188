"""The oracle's second source of random numbers (the restated generator of the HIP backend) is a valid
Sampler: analytic scenes stay exact, every stratum is visited once, and renders agree with the oracle's own
generator to Monte-Carlo error.  (GPU tests then compare device and oracle path by path on these numbers.)"""
import numpy as np

import scenes_util as S


def test_analytic_scenes_exact(O, cfg):
    rc = cfg.stratified_config(24, 24, 4, 4, min_depth=8, max_depth=12, seed=3)
    rgb, st = O.render_accumulate(S.furnace(), rc.render_desc(), device_numbers=True)
    assert np.allclose(rgb, np.array(S.L) * rc.spp, rtol=1e-12) and st["samples"] == 24 * 24 * 16
    albedo = np.array((0.8, 0.5, 0.25))
    rgb, _ = O.render_accumulate(S.lambert_floor(tuple(albedo)), rc.render_desc(), device_numbers=True)
    px = rgb / rc.spp
    floor = np.all(np.isclose(px, albedo * np.array(S.L), rtol=1e-9), axis=-1)
    sky = np.all(np.isclose(px, S.L, rtol=1e-9), axis=-1)
    assert floor.mean() > 0.5 and (floor | sky).mean() > 0.9


def test_random_sampler_traces_spp_plus_one(O, cfg):  # src/sampler.rs:48-55
    rc = cfg.random_config(8, 8, 10, seed=4)
    rgb, st = O.render_accumulate(S.furnace(), rc.render_desc(), device_numbers=True)
    assert st["samples"] == 8 * 8 * 11


def test_agrees_with_own_generator(O, cfg, cornell):
    rc = cfg.stratified_config(32, 32, 16, 16, seed=5)
    a, sa = O.render_accumulate(cornell, rc.render_desc(), device_numbers=True)
    b, sb = O.render_accumulate(cornell, rc.render_desc())
    assert abs(a.mean() - b.mean()) < 0.06 * b.mean()  # 262 k samples each, ~1.5 % noise on the mean
    assert abs(sa["rays"] / sa["samples"] - sb["rays"] / sb["samples"]) < 0.02 * sb["rays"] / sb["samples"]
    # first-bounce stratification shows: pixel (u, v) offsets cover the pixel evenly, so the direct view of
    # the light (radiance 100 per covered sample) has the same coverage with both generators
    lit_a, lit_b = (a[..., 1] > 50 * rc.spp).sum(), (b[..., 1] > 50 * rc.spp).sum()
    assert abs(int(lit_a) - int(lit_b)) <= 2


def test_reproduces_the_committed_renders(O, cfg, scenes):
    """tests/golden/paths.npz (tools/make_golden.py): the renders the GPU tests hold the device kernels to."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "paths.npz"))
    rc = cfg.stratified_config(32, 32, 4, 4, seed=61)
    for name, flat in scenes.items():
        for mode in ("f64", "f32"):
            rgb, st = O.render_accumulate(flat, rc.render_desc(), device_numbers=mode)
            assert np.allclose(rgb, g[f"{name}_{mode}"], rtol=1e-9, atol=1e-12)  # (libm may differ in the last bit)
            assert st["rays_live"] == int(g[f"{name}_{mode}_rays"])
