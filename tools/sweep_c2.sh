#!/bin/bash
# wavefront pipeline on the Cornell box: queue capacity sweep, and the fused kernel beside it
cat > /tmp/c2.py <<'PY'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, ROOT)
from rustraytracer_amd import api, config
be = api.Backend(0)
sc = api.Scene(config.load_scene(os.path.join(ROOT, "configs/cornell_box/scene.toml")))
rc = config.stratified_config(500, 500, 32, 32)
for pipe, name in ((api.PIPELINE_FUSED, "fused"), (api.PIPELINE_WAVEFRONT, "wavefront")):
    for rep in range(2):
        rgb, st = be.render_accumulate(sc, rc.render_desc(pipeline=pipe))
    print(name, "ms %.2f Msamples/s %.1f Mrays/s %.1f" % (st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3, st["rays"] / st["kernel_ms"] / 1e3), flush=True)
PY
for cap in 1048576 4194304 16777216; do echo "capacity=$cap"; RRT_DEBUG_WF_CAPACITY=$cap timeout -k 10 100 python3 -u /tmp/c2.py 2>&1 | grep -v amdgpu; done
