"""How far the f32 radiance accumulator carries (device only): a 4x4 image of the Cornell box at 2^16 .. 2^26 samples per
pixel in ONE call, against the same samples in 256 calls of a 256th each whose sums are added in f64 on the host."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustraytracer_amd import api, config  # noqa: E402

flat = config.load_scene(os.path.join(ROOT, "configs/cornell_box/scene.toml"))
be, sc = api.Backend(0), api.Scene(flat)
for log_side in (8, 10, 12, 13):
    side = 1 << log_side
    spp = side * side
    rc = config.stratified_config(4, 4, side, side, seed=2)
    parts = np.zeros((4, 4, 3), dtype=np.float64)
    n = 256
    for k in range(n):
        img, _ = be.render_accumulate(sc, rc.render_desc(sample_begin=spp * k // n, sample_end=spp * (k + 1) // n))
        parts += img
    line = ["2^%d spp" % (2 * log_side)]
    for name, pl in (("fused", api.PIPELINE_FUSED), ("wavefront", api.PIPELINE_WAVEFRONT)):
        img, _ = be.render_accumulate(sc, rc.render_desc(pipeline=pl))
        ok = parts > 0
        rel = np.abs(img[ok] - parts[ok]) / parts[ok]
        line.append("%s: one call off the f64 sum of 256 calls by %.1e (mean), %.1e (worst pixel)" % (name, rel.mean(), rel.max()))
    print("; ".join(line), flush=True)
