import os, sys, shutil, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "rustraytracer_amd", "csrc", "librrt_hip.so")
shutil.copy(lib, "/tmp/librrt_base.so")
for name in ["base"] + sys.argv[1:]:
    src = "/tmp/librrt_base.so" if name == "base" else os.path.join(ROOT, "expso", f"librrt_{name}.so")
    shutil.copy(src, lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--spp-side", "64"], capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        rays = j["mrays_per_s"] * j["ms_per_step"] * 1e3
        print(f"{name:10s} ms/step {j['ms_per_step']:.2f}  rays/sample {j['config']['rays_per_sample']:.3f}  ns per 64 rays {j['ms_per_step']*1e6/(rays/64)*7168/1:.1f} (per wave, 7168 waves)")
    except Exception as e:
        print(name, "failed", r.stdout[-300:], r.stderr[-300:])
shutil.copy("/tmp/librrt_base.so", lib)
