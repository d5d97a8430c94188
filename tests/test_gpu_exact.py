"""RRT_RENDER_EXACT_HITS: the f32 renderer with every closest-hit query tracked (a rounding-error bound for every
decision) and repeated in f64 where f32 cannot decide.  Same samples, same sampler, same paths as the plain f32 render
wherever f32 was right; the pixels that leave the oracle's paths can only become fewer.  The parity entry points
(rrt_intersect, rrt_trace_primary) always work this way in f32: tests/test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _off(dev, ref):
    err = np.abs(dev.astype(np.float64) - ref)
    return np.any(err > 2e-5 * np.abs(ref) + 1e-6, axis=-1)


@pytest.mark.parametrize("name", ["cornell_box", "smallpt"])
def test_exact_hits_render_follows_the_oracle(gpu, O, cfg, scenes, name):
    api, be = gpu
    sc = api.Scene(scenes[name])
    rc = cfg.stratified_config(96, 72, 8, 8, seed=41)
    ref, rst = O.render_accumulate(scenes[name], rc.render_desc(), device_numbers="f32")
    plain, st0 = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_FUSED))
    exact, st1 = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_FUSED, flags=api.RENDER_EXACT_HITS))
    assert st0["exact_redos"] == 0 and 0 < st1["exact_redos"] < st1["rays"]
    assert st1["samples"] == st0["samples"] == rst["samples"] and st1["nonfinite"] == 0
    off0, off1 = _off(plain, ref), _off(exact, ref)
    print(name, "pixels off the oracle's paths: plain f32", int(off0.sum()), "exact hits", int(off1.sum()), "of", off0.size,
          "; queries repeated in f64:", st1["exact_redos"], "of", st1["rays"])
    assert off1.sum() <= off0.sum() and off1.mean() <= 0.005
    assert abs(st1["rays"] - rst["rays_live"]) <= 1e-4 * rst["rays_live"]
    # through a tree: the per-lane walk tracks its leaf tests and the order of the hits it accepts
    tree, st2 = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_FUSED, accel=api.ACCEL_SAH_GPU, flags=api.RENDER_EXACT_HITS))
    assert st2["node_tests"] > 0 and st2["exact_redos"] > 0 and _off(tree, ref).sum() <= off0.sum()


def test_exact_hits_flag_rules(gpu, cfg, cornell):
    api, be = gpu
    sc = api.Scene(cornell)
    rc = cfg.stratified_config(32, 32, 2, 2, seed=3)
    for kwargs in (dict(pipeline=api.PIPELINE_WAVEFRONT, flags=api.RENDER_EXACT_HITS),
                   dict(flags=api.RENDER_EXACT_HITS | api.RENDER_NEE)):
        with pytest.raises(api.RrtError) as e:
            be.render_accumulate(sc, rc.render_desc(**kwargs))
        assert e.value.code == -3
    with pytest.raises(api.RrtError) as e:
        be.render_accumulate(sc, rc.render_desc(flags=4))
    assert e.value.code == -1
    # f64 has nothing to repeat: the flag is accepted and changes nothing
    a, sa = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F64))
    b, sb = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F64, flags=api.RENDER_EXACT_HITS))
    assert np.allclose(a, b, rtol=1e-6) and sb["exact_redos"] == 0 and sa["rays"] == sb["rays"]
