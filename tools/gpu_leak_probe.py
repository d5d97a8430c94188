"""Device memory after many create / render / destroy cycles (device only): scenes with and without trees, both
pipelines, both precisions, next-event estimation; free device memory must come back to where it started."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from rustraytracer_amd import _abi, api, config  # noqa: E402
import test_gpu_lbvh as T  # noqa: E402

cornell = config.load_scene(os.path.join(ROOT, "configs/cornell_box/scene.toml"))
mesh = T.mesh_scene(50000)
rc = config.stratified_config(128, 96, 2, 2)


def cycle():
    for flat in (cornell, mesh):
        be, sc = api.Backend(0), api.Scene(flat)
        for accel in (api.ACCEL_AUTO, api.ACCEL_LBVH, api.ACCEL_SAH, api.ACCEL_SAH_GPU):
            for pl in (api.PIPELINE_FUSED, api.PIPELINE_WAVEFRONT):
                for prec in (api.PRECISION_F32, api.PRECISION_F64):
                    be.render_accumulate(sc, rc.render_desc(accel=accel, pipeline=pl, precision=prec))
        if flat is cornell:
            be.render_accumulate(sc, rc.render_desc(flags=_abi.RENDER_NEE, pipeline=api.PIPELINE_WAVEFRONT, accel=api.ACCEL_SAH_GPU))
            be.render(sc, rc)
        del sc, be


cycle()
torch.cuda.synchronize()
free0, _ = torch.cuda.mem_get_info()
for k in range(20):
    cycle()
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
print("free device memory before %d MiB, after 20 cycles %d MiB, difference %.2f MiB" % (free0 >> 20, free1 >> 20, (free0 - free1) / 2**20))
sys.exit(0 if free0 - free1 < (8 << 20) else 1)
