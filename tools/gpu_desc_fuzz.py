"""Random render descriptions through rrt_render_accumulate (device only): gpu_desc_fuzz.py [calls, default 3000] [seed]
Fields drawn from valid and invalid values alike (zero and odd sizes, strata that do not multiply to the sample count,
empty and reversed sample ranges, depth limits 0 .. 600, unknown enum values, stray flag bits, every flag combination).
A call either fails with an error code and a message, or renders: exact sample count for its range, finite sums.
Never a crash, a hang or a device fault."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rustraytracer_amd import _abi, api, config  # noqa: E402
import scenes_util as S  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    be = api.Backend(0)
    scenes = [api.Scene(config.load_scene(os.path.join(ROOT, "configs/cornell_box/scene.toml"))),
              api.Scene(config.load_scene(os.path.join(ROOT, "configs/smallpt/scene.toml"))), api.Scene(S.all_prims_scene())]
    ok = err = bad = 0
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    for it in range(n):
        # a valid description first ...
        d = _abi.RenderDesc()
        d.width, d.height = pick([1, 2, 3, 17, 64, 65]), pick([1, 2, 5, 48, 64])
        d.sampler_type = pick([0, 1, 1, 1])
        xs, ys = pick([1, 2, 3, 4, 8]), pick([1, 2, 3, 5, 8])
        d.x_strata, d.y_strata, d.spp = xs, ys, xs * ys
        d.min_depth, d.max_depth = pick([0, 1, 2, 3, 10, 700]), pick([0, 1, 2, 6, 6, 6, 20, 511])
        d.sample_begin = pick([0, 0, 0, 1, d.spp - 1])
        d.sample_end = pick([d.spp, d.spp, d.sample_begin + 1]) + (1 if d.sampler_type == 0 and rng.random() < 0.5 else 0)
        d.seed = int(rng.integers(0, 2**63))
        d.precision, d.accel, d.pipeline = pick([0, 0, 1]), pick([0, 1, 2, 3, 4]), pick([0, 1, 2])
        d.flags = pick([0, 0, 1, 2])
        # ... then up to two fields spoilt
        for _ in range(pick([0, 0, 1, 1, 2])):
            name = pick(["width", "height", "sampler_type", "spp", "x_strata", "y_strata", "max_depth", "sample_begin", "sample_end",
                         "precision", "accel", "pipeline", "flags"])
            setattr(d, name, pick({"width": [0, 2**31], "height": [0, 2**30], "sampler_type": [2, 7], "spp": [0, d.spp + 1, 7],
                                   "x_strata": [0, 5], "y_strata": [0, 7], "max_depth": [512, 600, 2**31], "sample_begin": [d.sample_end, 2**31, 2**40],
                                   "sample_end": [0, d.sample_begin, 2**31 + 1, d.spp + 5], "precision": [2, 9], "accel": [5, 99], "pipeline": [3, 8],
                                   "flags": [3, 4, 0x80, 0xffffffff]}[name]))
        sc = pick(scenes)
        acc = np.zeros((min(max(d.height, 1), 4096), min(max(d.width, 1), 4096), 3), dtype=np.float32)  # (a description that asks for more is refused before the sums are touched)
        st = _abi.Stats()
        import ctypes as C
        rc = be._lib.rrt_render_accumulate(be.handle, sc.handle, C.byref(d), acc.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))
        if rc == 0:
            ok += 1
            expect = d.width * d.height * (d.sample_end - d.sample_begin)
            if st.samples != expect or not np.all(np.isfinite(acc)) or st.stack_overflows:
                bad += 1
                print("call %d rendered %d samples, expected %d, finite %s" % (it, st.samples, expect, bool(np.all(np.isfinite(acc)))), flush=True)
        else:
            err += 1
            msg = be._lib.rrt_last_error()
            if rc > 0 or not msg:
                bad += 1
                print("call %d: status %d without a message" % (it, rc), flush=True)
    print("calls %d: rendered %d, refused %d, problems %d" % (n, ok, err, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
