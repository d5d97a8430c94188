#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by tools/profile_r01.sh) into profiles/:
   <tag>_kernel_stats.csv   the rocprofv3 --kernel-trace --stats table
   <tag>_pmc.csv            per-dispatch counter values of the render kernel, all passes
   <tag>_summary.md         derived figures
   hbm_traffic.json         bytes per launch of the render kernel, read by bench.py"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
kern = [r for r in rows if "render_fused" in r["Name"]][0]
avg_ms = float(kern["AverageNs"]) / 1e6

vals = collections.defaultdict(list)
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["pass", "dispatch", "kernel", "counter", "value"])
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True)):
        pas = f.split(os.sep + "pmc_")[1].split(os.sep)[0]
        for r in csv.DictReader(open(f)):
            if "render_fused" in r["Kernel_Name"]:
                w.writerow([pas, r["Dispatch_Id"], "render_fused<float,false>", r["Counter_Name"], r["Counter_Value"]])
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in vals.items()}
bench = {}
for f in glob.glob(os.path.join(src, "trace.log")):
    for line in open(f):
        if line.startswith("{"):
            bench = json.loads(line)
fetch_b = c.get("FETCH_SIZE", 0.0) * 1024 * 2  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
write_b = c.get("WRITE_SIZE", 0.0) * 1024
json.dump({"bytes_per_launch": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
           "source": f"profiles/{tag}_pmc.csv", "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes"},
          open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
clock = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (avg_ms * 1e-3) / 1e9 if avg_ms else 0
rays = bench.get("mrays_per_s", 0) * bench.get("ms_per_step", 0) * 1e3
iters = rays / 64 if rays else float("nan")
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}`: `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`\n\n")
    f.write(f"* kernel `render_fused<float,false>`: {kern['Calls']} calls, average {avg_ms:.3f} ms "
            f"(min {float(kern['MinNs'])/1e6:.3f}, max {float(kern['MaxNs'])/1e6:.3f}), {kern['Percentage']} % of GPU time\n")
    if bench:
        r = bench["roofline"]
        f.write(f"* bench line of the traced run: {bench['value']:.0f} Msamples/s, {bench['mrays_per_s']:.0f} Mrays/s, "
                f"{bench['ms_per_step']:.2f} ms/step; roofline achieved {r['achieved']:.2f} TFLOP/s "
                f"({100*r['frac']:.1f} % of {r['peak']} FP32 VALU peak), avg launch {r['avg_launch_ms']:.3f} ms (HIP events)\n")
    f.write(f"* effective clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time): {clock:.2f} GHz\n")
    if "SQ_INSTS_VALU" in c:
        f.write(f"* per launch: SQ_INSTS_VALU {c['SQ_INSTS_VALU']:.4g}, SQ_INSTS_SALU {c['SQ_INSTS_SALU']:.4g}, "
                f"SQ_INSTS_SMEM {c['SQ_INSTS_SMEM']:.4g}, SQ_WAVES {c['SQ_WAVES']:.0f}\n")
        f.write(f"* per wave trip of the main loop (64 rays): VALU {c['SQ_INSTS_VALU']/iters:.0f}, "
                f"SALU {c['SQ_INSTS_SALU']/iters:.0f}, SMEM {c['SQ_INSTS_SMEM']/iters:.1f}\n")
        f.write(f"* SIMD lane efficiency (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64): "
                f"{c['SQ_THREAD_CYCLES_VALU']/c['SQ_ACTIVE_INST_VALU']/64:.3f}\n")
        f.write(f"* VALU issue interval: one wave64 VALU instruction per "
                f"{1024*avg_ms*1e-3*clock*1e9/c['SQ_INSTS_VALU']:.2f} cycles per SIMD (tools/ubench/valu_rate.hip measures "
                f"~4 cycles per instruction for back-to-back v_fma_f32 / v_pk_fma_f32: the kernel is VALU-issue bound)\n")
    if "SQ_WAIT_ANY" in c:
        wc = c["SQ_WAVE_CYCLES"]
        f.write(f"* wave-cycle shares: waiting on s_waitcnt {c['SQ_WAIT_ANY']/wc:.2f}, issue stalls "
                f"{c['SQ_WAIT_INST_ANY']/wc:.2f}, issuing {c['SQ_ACTIVE_INST_ANY']/wc:.2f}\n")
    f.write(f"* HBM traffic per launch: FETCH_SIZE {c.get('FETCH_SIZE',0):.0f} KiB (x2 on gfx950 = {fetch_b/1e6:.2f} MB), "
            f"WRITE_SIZE {c.get('WRITE_SIZE',0):.4g} KiB = {write_b/1e9:.2f} GB (the per-item float atomics into the accumulator, "
            f"one 64-B fabric request per lane) -> {(fetch_b+write_b)/(avg_ms*1e-3)/1e9:.1f} GB/s\n")
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
