import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "torch_first":
    import torch
    print("torch first: cuda available", torch.cuda.is_available())
    x = torch.zeros(4, device="cuda")
    from rustraytracer_amd import api
    be = api.Backend(0)
    print("then lib ok")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "lib_first":
    from rustraytracer_amd import api
    be = api.Backend(0)
    import torch
    print("lib first: torch cuda available", torch.cuda.is_available())
    try:
        x = torch.zeros(4, device="cuda"); print("torch alloc ok")
    except Exception as e:
        print("torch alloc failed:", e)
    sys.exit(0)
for mode in ("torch_first", "lib_first"):
    r = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True)
    print(mode, "->", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
from rustraytracer_amd import api, config
from oracle import oracle_py as O
import scenes_util as S
be = api.Backend(0)
rc = config.stratified_config(48, 48, 4, 4, min_depth=8, max_depth=12)
for prec in (0, 1):
    rgb, st = be.render_accumulate(api.Scene(S.glass_ball()), rc.render_desc(precision=prec))
    frac = (rgb / rc.spp) / np.array(S.L)
    print("glass prec", prec, "max", frac.max(), "mean", frac.mean(), "n>1", int((frac > 1.0001).sum()), st)
    ys, xs = np.nonzero(frac[..., 0] > 1.0001)
    print("   where", list(zip(ys[:10], xs[:10])), frac[ys[:5], xs[:5], 0])
ref, rst = O.render_accumulate(S.glass_ball(), rc.render_desc())
frac = (ref / rc.spp) / np.array(S.L)
print("oracle glass max", frac.max(), "mean", frac.mean(), rst)
