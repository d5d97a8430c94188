"""The benchmark configurations at their FULL sizes, checked through properties that do not depend on
the image size (the oracle cannot render 4 G samples in a test):

* pixel x of a W-wide image integrates u over [(x + 0.5) / W, (x + 1.5) / W] (src/renderer.rs:120-121:
  the sampler's offset in [0, 1) is added to x + 0.5, so the footprint sits half a pixel to the right
  of and below the pixel).  Pixel x of a k-times smaller image therefore covers the full-size pixels
  k x + k / 2 ... + k - 1 exactly, when k is odd; for an even k the small render's camera is moved by
  half a FULL-size pixel, which makes it so.  The full-size render averaged over these blocks and the
  small render then estimate the same integrals (the small image's last row and column see part of
  the image plane the full-size image does not cover and are left out).
  The small renders come from the f64 oracle (two seeds); the full-size GPU image must be
  closer to either of them than they are to each other, in LINEAR radiance (the oracle's 1024-spp
  pixels are sums of a handful of light hits; a concave tone curve would bias them low), and cell
  means on a coarse grid must agree to Monte-Carlo error;
* the two halves of the sample range add up to the whole render (the multi-GPU decomposition);
* counters: every (pixel, sample) was traced exactly once and nothing non-finite was produced.

Sizes: Cornell box 500x500 @ 16384 (src: configs/cornell_box/render_high_quality.toml), smallpt
1024x768 @ 16384, Cornell + cube 1024x1024 @ 4096, million-triangle scene 3840x2160 @ 1024.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# name: (scene, width, height, strata per side, downsampling factor, small pixels per side of a coarse cell)
FULL = {
    "cornell": ("configs/cornell_box/scene.toml", 500, 500, 128, 10, 10),
    "smallpt": ("configs/smallpt/scene.toml", 1024, 768, 128, 16, 16),
    "cornell_cube": ("configs/cornell_cube/scene.toml", 1024, 1024, 64, 16, 16),
}


def box_down(img, k):
    h, w, c = img.shape
    h, w = h // k * k, w // k * k
    return img[:h, :w].reshape(h // k, k, w // k, k, c).mean(axis=(1, 3))


def resample(img, k):
    """Means of the full-size pixels inside the footprints of the small image's pixels (all but the last
    row and column)."""
    h, w = img.shape[0] // k - 1, img.shape[1] // k - 1
    return box_down(img[k // 2:k // 2 + h * k, k // 2:k // 2 + w * k], k)


def small_scene(cfg, flat, W, H, k):
    """The scene whose W/k x H/k render has pixel footprints made of whole full-size pixels."""
    cam = dict(flat.camera)
    if k % 2 == 0:
        cam["lower_left_corner"] = (np.asarray(cam["lower_left_corner"], dtype=np.float64)
                                    + 0.5 / W * np.asarray(cam["horizontal"], dtype=np.float64)
                                    - 0.5 / H * np.asarray(cam["vertical"], dtype=np.float64))
    return cfg.FlatScene(cam, flat.prims, flat.materials)


def rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_size_against_oracle_at_reduced_size(gpu, O, cfg, name):
    api, be = gpu
    path, W, H, side, k, coarse = FULL[name]
    flat = cfg.load_scene(os.path.join(ROOT, path))
    sc = api.Scene(flat)
    rc = cfg.stratified_config(W, H, side, side, seed=31)
    full, st = be.render_accumulate(sc, rc.render_desc())
    assert st["samples"] == W * H * side * side and st["nonfinite"] == 0
    assert st["rays"] > 2 * st["samples"]  # closed scenes: a path makes more than two queries on average

    # halves of the sample range (what two GPUs would trace) add up to the whole
    acc = np.zeros_like(full)
    rays = 0
    for lo, hi in ((0, rc.spp // 2), (rc.spp // 2, rc.spp)):
        _, s2 = be.render_accumulate(sc, rc.render_desc(sample_begin=lo, sample_end=hi), acc)
        rays += s2["rays"]
    assert rays == st["rays"]  # same paths
    assert np.allclose(acc, full, rtol=1e-4, atol=1e-2)  # float atomics: only the summation order differs

    # oracle at 1/k of the size, 1024 spp, two seeds
    moved = small_scene(cfg, flat, W, H, k)
    small = [O.render_accumulate(moved, cfg.stratified_config(W // k, H // k, 32, 32, seed=s).render_desc(),
                                 n_threads=min(16, os.cpu_count() or 1))[0] / 1024.0 for s in (41, 42)]
    down = resample(full.astype(np.float64) / rc.spp, k)
    check_against_small_renders(down, [x[:-1, :-1] for x in small], coarse)


def check_against_small_renders(down, small, coarse):
    base = rmse(small[0], small[1])  # sqrt(2) x the noise of one small render
    cd, cm = box_down(down, coarse), box_down(0.5 * (small[0] + small[1]), coarse)
    print("rmse ratios", [rmse(down, ref) / base for ref in small], "worst cell", float(np.max(np.abs(cd - cm) / (cm + 1e-3))),
          "mean ratio", cd.mean() / cm.mean())
    for ref in small:
        # the full-size image carries next to no noise of its own, so the distance is ~ base / sqrt(2)
        assert rmse(down, ref) <= 0.85 * base
    # cells of coarse x coarse small pixels: >= 100 pixels x 2 renders x 1024 spp each, a few per cent of noise
    assert np.all(np.abs(cd - cm) <= 0.1 * cm + 1e-3), float(np.max(np.abs(cd - cm) / (cm + 1e-3)))
    assert abs(cd.mean() - cm.mean()) <= 0.02 * cm.mean()


def test_million_triangle_scene_full_size(gpu, cfg):
    """3840x2160 @ 1024 through the tree and the wavefront pipeline (what AUTO picks).  The oracle scans a
    million triangles per ray, so the small renders come from the backend itself, whose closest hit on
    this scene is pinned against the linear scan and the oracle in test_gpu_lbvh.py."""
    import make_c5_assets
    api, be = gpu
    make_c5_assets.main()
    flat = cfg.load_scene(os.path.join(ROOT, "configs", "mesh_1m", "scene.toml"))
    sc = api.Scene(flat)
    W, H, side, k = 3840, 2160, 32, 8
    rc = cfg.stratified_config(W, H, side, side, seed=31)
    import torch
    dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")  # 99.5 MB stay on the device
    _, st = be.render_accumulate(sc, rc.render_desc(), dev)
    assert st["samples"] == W * H * side * side and st["nonfinite"] == 0
    assert st["node_tests"] > st["rays"] and st["prim_tests"] < 40 * st["rays"]  # the tree ran, not the scan
    down = resample(dev.cpu().numpy().astype(np.float64) / rc.spp, k)
    sc_small = api.Scene(small_scene(cfg, flat, W, H, k))
    small = [be.render_accumulate(sc_small, cfg.stratified_config(W // k, H // k, 32, 32, seed=s).render_desc())[0]
             .astype(np.float64) / 1024.0 for s in (41, 42)]
    check_against_small_renders(down, [x[:-1, :-1] for x in small], 30)
