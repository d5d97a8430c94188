"""Quick GPU bring-up check: parity of primary hits, a small render vs the oracle, a timing run."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustraytracer_amd import api, config
from oracle import oracle_py as O

be = api.Backend(0)
for name in ("cornell_box", "smallpt"):
    flat = config.load_scene(os.path.join(ROOT, "configs", name, "scene.toml"))
    sc = api.Scene(flat)
    rng = np.random.default_rng(7)
    u, v = rng.random(100000), rng.random(100000)
    ref = O.trace_primary(flat, u, v)
    for prec in (api.PRECISION_F64, api.PRECISION_F32):
        oid, pid, t = be.trace_primary(sc, u, v, precision=prec)
        hit = ref[0] >= 0
        bad = int(np.sum((oid != ref[0]) | (pid != ref[1])))
        rel = np.max(np.abs(t[hit & (oid == ref[0])] - ref[2][hit & (oid == ref[0])]) / ref[2][hit & (oid == ref[0])])
        print(f"{name} prec={prec}: id mismatches {bad}/100000, max rel dt {rel:.3e}, misses {int((~hit).sum())}")
    rc = config.stratified_config(128, 128, 16, 16)
    for prec in (api.PRECISION_F32, api.PRECISION_F64):
        t0 = time.time()
        rgb, st = be.render_accumulate(sc, rc.render_desc(precision=prec))
        dt = time.time() - t0
        print(f"{name} render prec={prec}: {st} wall {dt:.3f}s")
        ref_rgb, ref_st = O.render_accumulate(flat, rc.render_desc(seed=5))
        print("   mean gpu %.5f oracle %.5f  rays/sample gpu %.4f oracle %.4f" % (
            rgb.mean() / rc.spp, ref_rgb.mean() / rc.spp, st["rays"] / st["samples"], ref_st["rays"] / ref_st["samples"]))
        img = be.postprocess(rgb, rc.spp, rc.post_desc())
        oimg, _ = O.postprocess(ref_rgb, rc.spp, rc.post_desc())
        print("   u8 mean gpu %.3f oracle %.3f" % (img.mean(), oimg.mean()))
        if prec == api.PRECISION_F32:
            from PIL import Image
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            Image.fromarray(img).save(os.path.join(ROOT, "gpurun_out", f"{name}_gpu.png"))
    # throughput at a bigger size
    rc = config.stratified_config(500, 500, 32, 32)
    for it in range(2):
        rgb, st = be.render_accumulate(sc, rc.render_desc())
        print(f"{name} 500x500@1024: kernel {st['kernel_ms']:.2f} ms, {st['samples']/st['kernel_ms']/1e3:.1f} Msamples/s, "
              f"{st['rays']/st['kernel_ms']/1e3:.1f} Mrays/s, rays/sample {st['rays']/st['samples']:.3f}")
