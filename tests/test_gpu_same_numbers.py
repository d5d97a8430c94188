"""Every f32 kernel variant computes the same numbers for the same ray (the library is built with -ffp-contract=on, so a
multiply-add is fused where the source writes one and nowhere else).  Random scenes on which the wavefront pipeline's two
shade kernels -- the one that scans, the one that reads the tree kernel's hits -- used to differ in the last bit of some
throughput or direction and then lost a path or two: the scan and the tree must give equal ray and bounce counts, and so
must the two pipelines."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_scan_tree_and_both_pipelines_trace_the_same_paths_in_f32(gpu, cfg):
    api, be = gpu
    from rustraytracer_amd import _abi
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_exact_fuzz as F
    rng = np.random.default_rng(3)
    wanted = {113, 122, 132, 135, 137, 140, 142, 149}  # (differed before the flag)
    checked = 0
    for s in range(150):
        flat, _, _, _ = F.random_scene(rng)
        m = np.zeros(4, dtype=_abi.MATERIAL_DTYPE)
        for i, (typ, rgb, param) in enumerate(((_abi.MAT_EMISSIVE, (2.0, 1.5, 1.0), 0.0), (_abi.MAT_LAMBERTIAN, (0.7, 0.6, 0.5), 0.0),
                                               (_abi.MAT_REFLECTOR, (0, 0, 0), 0.0), (_abi.MAT_DIELECTRIC, (0, 0, 0), 1.5))):
            m["type"][i], m["rgb"][i], m["param"][i] = typ, rgb, param
        flat.prims["object_id"] = rng.integers(0, 4, flat.n_prims)
        if s not in wanted:
            continue
        sc = api.Scene(cfg.FlatScene(flat.camera, flat.prims, m))
        rc = cfg.stratified_config(64, 48, 2, 2, min_depth=1, max_depth=6, seed=s)
        keys = {}
        for pl in (api.PIPELINE_FUSED, api.PIPELINE_WAVEFRONT):
            for accel in (api.ACCEL_LINEAR, api.ACCEL_SAH_GPU):
                _, st = be.render_accumulate(sc, rc.render_desc(pipeline=pl, accel=accel))
                keys[(pl, accel)] = (st["rays"], tuple(st["bounces"]), st["samples"])
        assert len(set(keys.values())) == 1, (s, keys)
        checked += 1
    assert checked == len(wanted)
