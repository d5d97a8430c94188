"""Wavefront pipeline (HBM ray/hit queues, per-bounce kernels) against the fused pipeline: both draw every
path from the same counter-based sampler, so they trace the SAME paths -- ray, bounce and sample counts
must be equal and the radiance sums may differ only by float summation order -- and against the oracle
on the analytic scenes."""
import os

import numpy as np
import pytest

import scenes_util as S

pytestmark = pytest.mark.gpu


def both(api, be, sc, rc, accel):
    out = []
    for pipe in (api.PIPELINE_FUSED, api.PIPELINE_WAVEFRONT):
        out.append(be.render_accumulate(sc, rc.render_desc(accel=accel, pipeline=pipe)))
    return out


def same_paths(a, b):
    (rgb_a, st_a), (rgb_b, st_b) = a, b
    for k in ("samples", "rays", "prim_tests", "node_tests", "nonfinite"):
        assert st_a[k] == st_b[k], k
    assert list(st_a["bounces"]) == list(st_b["bounces"])
    scale = max(float(np.abs(rgb_a).max()), 1e-30)
    assert np.abs(rgb_a - rgb_b).max() <= 4e-6 * scale


@pytest.mark.parametrize("name", ["cornell_box", "smallpt"])
def test_same_paths_as_fused_linear(gpu, cfg, scenes, name):
    api, be = gpu
    sc = api.Scene(scenes[name])
    for rc in (cfg.stratified_config(64, 48, 4, 4), cfg.random_config(37, 21, 9, seed=3),
               cfg.stratified_config(7, 5, 2, 3)):  # the last one: fewer pixels than a wave has lanes
        same_paths(*both(api, be, sc, rc, api.ACCEL_LINEAR))


def test_same_paths_as_fused_lbvh(gpu, cfg, cornell):
    api, be = gpu
    sc = api.Scene(cornell)
    for tree in (api.ACCEL_LBVH, api.ACCEL_SAH, api.ACCEL_SAH_GPU):
        same_paths(*both(api, be, sc, cfg.stratified_config(96, 96, 4, 4), tree))
    # all primitive types and all materials, planes beside the tree
    sc = api.Scene(S.all_prims_scene())
    for tree in (api.ACCEL_LBVH, api.ACCEL_SAH, api.ACCEL_SAH_GPU):
        same_paths(*both(api, be, sc, cfg.stratified_config(80, 60, 4, 4), tree))


def test_queue_smaller_than_the_render(gpu, cfg, cornell, monkeypatch):
    """Many trips with the queue refilled from the sample cursors, and a drain at the end."""
    api, be = gpu
    sc = api.Scene(cornell)
    rc = cfg.stratified_config(100, 100, 8, 8)  # 640 k samples
    ref = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_FUSED))
    monkeypatch.setenv("RRT_DEBUG_WF_CAPACITY", "8192")
    small = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_WAVEFRONT))
    monkeypatch.delenv("RRT_DEBUG_WF_CAPACITY")
    big = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_WAVEFRONT))
    same_paths(ref, small)
    same_paths(ref, big)


def test_furnace_and_shards(gpu, cfg, cornell):
    api, be = gpu
    sc = api.Scene(S.furnace())
    rc = cfg.stratified_config(33, 17, 4, 4)
    rgb, st = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_WAVEFRONT))
    assert st["samples"] == 33 * 17 * 16 and st["rays"] == st["samples"] and sum(st["bounces"]) == 0
    assert np.array_equal(rgb, np.broadcast_to(np.float32(S.L) * 16, rgb.shape))
    # sample-index shards add up to the whole render
    sc = api.Scene(cornell)
    rc = cfg.stratified_config(40, 40, 8, 8, seed=5)
    full, st_full = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_WAVEFRONT))
    acc = np.zeros_like(full)
    rays = 0
    for lo, hi in ((0, 16), (16, 17), (17, 64)):
        _, st = be.render_accumulate(sc, rc.render_desc(sample_begin=lo, sample_end=hi, pipeline=api.PIPELINE_WAVEFRONT), acc)
        rays += st["rays"]
    assert rays == st_full["rays"]
    assert np.allclose(acc, full, rtol=2e-6, atol=1e-6)


def test_wavefront_rejects_what_it_cannot_do(gpu, cfg, cornell):
    api, be = gpu
    sc = api.Scene(cornell)
    rc = cfg.stratified_config(8, 8, 2, 2, max_depth=600)
    with pytest.raises(api.RrtError):
        be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_WAVEFRONT))
    # max_depth 0: nothing is traced (src/tracers/utils.rs:35 loop runs zero times)
    rc = cfg.stratified_config(8, 8, 2, 2, max_depth=0)
    rgb, st = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_WAVEFRONT))
    assert st["rays"] == 0 and not rgb.any()


@pytest.mark.parametrize("shape", [(1, 1, 256, 256), (4096, 2048, 1, 1), (3, 1000, 2, 2), (1000, 3, 1, 3)])
def test_extreme_image_shapes_both_pipelines(gpu, cfg, shape):
    """One pixel with 65536 samples, eight million pixels with one sample, images one wave wide or tall: every
    (pixel, sample) is traced exactly once by either pipeline (furnace scene: every sample returns exactly L)."""
    import scenes_util as S
    api, be = gpu
    w, h, xs, ys = shape
    sc = api.Scene(S.furnace())
    rc = cfg.stratified_config(w, h, xs, ys)
    for pipeline in (api.PIPELINE_FUSED, api.PIPELINE_WAVEFRONT):
        rgb, st = be.render_accumulate(sc, rc.render_desc(pipeline=pipeline))
        assert st["samples"] == w * h * xs * ys and st["rays"] == st["samples"] and st["nonfinite"] == 0
        want = np.float64(S.L) * (xs * ys)
        assert np.allclose(rgb, np.broadcast_to(want, rgb.shape), rtol=1e-6)  # sums of up to 512 chunk sums in f32
