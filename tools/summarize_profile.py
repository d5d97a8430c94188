#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_run.sh) into profiles/:
   <tag>_kernel_stats.csv   the rocprofv3 --kernel-trace --stats table
   <tag>_pmc.csv            counter values per kernel: sum over the run's dispatches, all passes
   <tag>_summary.md         derived figures per kernel
   hbm_traffic.json         HBM-side bytes per ray of the profiled workload (entry added or replaced), read by bench.py"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
command = open(os.path.join(src, "command.txt")).read().strip()


def short(name):
    m = re.search(r"rrt::(\w+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def newest(pattern):
    """One file per directory: a profile repeated under the same tag leaves the earlier run's files beside the new ones
    (results are merged into the local copy of the directory, never removed from it)."""
    by_dir = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


stats = newest(os.path.join(src, "trace", "**", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
kern = {}
for r in csv.DictReader(open(stats)):
    if "rrt::" in r["Name"]:
        kern[short(r["Name"])] = r

sums = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(int))
for f in newest(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "rrt::" in r["Kernel_Name"]:
            k = short(r["Kernel_Name"])
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]] += 1
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "counter", "sum_over_dispatches", "dispatches"])
    for k in sorted(sums):
        for c in sorted(sums[k]):
            w.writerow([k, c, repr(sums[k][c]), disp[k][c]])

bench = {}
for line in open(os.path.join(src, "trace.log")):
    if line.startswith("{"):
        bench = json.loads(line)

lines = [f"# rocprofv3 summary `{tag}`: `{command}`\n"]
if bench:
    r = bench["roofline"]
    lines.append(f"* bench line of the traced run: {bench['value']:.0f} Msamples/s, {bench['mrays_per_s']:.0f} Mrays/s, "
                 f"{bench['ms_per_step']:.2f} ms/step; roofline achieved {r['achieved']:.2f} TFLOP/s "
                 f"({100*r['frac']:.1f} % of {r['peak']} FP32 VALU peak), kernels {r['avg_launch_ms']:.3f} ms per step (HIP events)")
    if "queue_traffic_algorithmic_GBps" in r:
        lines.append(f"* algorithmic queue traffic (88 B per ray): {r['queue_traffic_algorithmic_GBps']:.0f} GB/s")
steps_total = (bench.get("steps", 0) + bench.get("warmup", 0)) or 1
hbm_total, hbm_disp = [0.0, 0.0], [0]
for k in sorted(kern, key=lambda k: -float(kern[k]["TotalDurationNs"])):
    r = kern[k]
    if float(r["Percentage"]) < 0.5:
        continue
    total_ms = float(r["TotalDurationNs"]) / 1e6
    lines.append(f"\n## `{k}`: {r['Calls']} calls, {total_ms:.2f} ms in total ({r['Percentage']} % of GPU time), "
                 f"average {float(r['AverageNs'])/1e6:.3f} ms (min {float(r['MinNs'])/1e6:.3f}, max {float(r['MaxNs'])/1e6:.3f})")
    c = sums.get(k, {})
    if not c:
        continue
    total_s = total_ms * 1e-3
    if "GRBM_GUI_ACTIVE" in c:
        clock = c["GRBM_GUI_ACTIVE"] / 8 / total_s / 1e9
        lines.append(f"* effective clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time): {clock:.2f} GHz")
    else:
        clock = 2.4
    if "SQ_INSTS_VALU" in c:
        lines.append(f"* SQ_WAVES {c['SQ_WAVES']:.4g}, SQ_INSTS_VALU {c['SQ_INSTS_VALU']:.4g}, SQ_INSTS_SALU {c['SQ_INSTS_SALU']:.4g}, "
                     f"SQ_INSTS_SMEM {c['SQ_INSTS_SMEM']:.4g} (sums over the run)")
        lines.append(f"* SIMD lane efficiency (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64): "
                     f"{c['SQ_THREAD_CYCLES_VALU']/c['SQ_ACTIVE_INST_VALU']/64:.3f}")
        per = 1024 * total_s * clock * 1e9 / c['SQ_INSTS_VALU']
        lines.append(f"* VALU issue: one wave64 VALU instruction per {per:.2f} cycles per SIMD; a full wave issues back to back "
                     f"every 4 cycles (tools/ubench/valu_rate.hip) -> {4 / per:.2f} instructions per 4-cycle slot"
                     + (" (above 1: integer shift/xor/add instructions issue every ~3.6 cycles, tools/ubench/valu_rate.hip; the EXEC mask makes no difference, tools/ubench/exec_rate.hip)" if per < 4 else ""))
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        lines.append(f"* wave-cycle shares: waiting on s_waitcnt {c['SQ_WAIT_ANY']/wc:.2f}, issue stalls "
                     f"{c['SQ_WAIT_INST_ANY']/wc:.2f}, issuing {c['SQ_ACTIVE_INST_ANY']/wc:.2f}; "
                     f"VMEM reads {c['SQ_INSTS_VMEM_RD']:.4g}, VMEM writes {c['SQ_INSTS_VMEM_WR']:.4g}, LDS {c['SQ_INSTS_LDS']:.4g} wave instructions")
    if "TCC_REQ_sum" in c:
        lines.append(f"* L2: {c['TCC_REQ_sum']:.4g} requests, hit rate {c['TCC_HIT_sum']/max(c['TCC_HIT_sum']+c['TCC_MISS_sum'],1):.2f}")
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        fetch_b = c.get("FETCH_SIZE", 0.0) * 1024 * 2  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
        write_b = c.get("WRITE_SIZE", 0.0) * 1024
        lines.append(f"* HBM-side traffic over the run: FETCH_SIZE x2 (gfx950 correction) = {fetch_b/1e9:.3f} GB read, WRITE_SIZE = "
                     f"{write_b/1e9:.3f} GB written -> {(fetch_b+write_b)/total_s/1e9:.1f} GB/s while the kernel runs")
        hbm_total[0] += fetch_b
        hbm_total[1] += write_b
        hbm_disp[0] = max(hbm_disp[0], disp[k].get("FETCH_SIZE", 0), disp[k].get("WRITE_SIZE", 0))
# HBM-side bytes per ray of this workload (all kernels of the run), kept for bench.py's roofline.traffic
if bench and (hbm_total[0] or hbm_total[1]):
    wl = bench.get("config", {}).get("workload", "")
    name = {"configs/cornell_box": "cornell", "configs/smallpt": "smallpt", "configs/cornell_cube": "cornell_cube",
            "configs/mesh_1m": "mesh_1m"}.get(wl.split("/scene.toml")[0], None)
    if name:
        if bench["config"]["pipeline"].startswith("wavefront") and name != "mesh_1m":
            name += "_wavefront"
        if bench.get("dtype") == "f64":
            name += "_f64"
        # rays of the profiled run: every launch of the run (warm-up + steps) traces the same rays
        rays = bench["mrays_per_s"] * 1e6 * bench["ms_per_step"] * 1e-3 * (bench["steps"] + bench["warmup"])
        path = os.path.join(dst, "hbm_traffic.json")
        try:
            table = json.load(open(path))
        except Exception:
            table = {}
        if "workloads" not in table:
            table = {"workloads": {}}
        table["note"] = ("HBM-side bytes per ray from separate rocprofv3 --pmc passes (FETCH_SIZE x2, the gfx950 correction, "
                         "+ WRITE_SIZE; KiB -> bytes), summed over all kernels of the profiled bench.py run and divided by the "
                         "rays that run traced; bench.py multiplies by the rays of its own step")
        table["workloads"][name] = {"bytes_per_ray": (hbm_total[0] + hbm_total[1]) / rays, "fetch_bytes_per_ray": hbm_total[0] / rays,
                                    "write_bytes_per_ray": hbm_total[1] / rays, "rays_profiled": rays,
                                    "source": f"profiles/{tag}_pmc.csv", "command": command}
        json.dump(table, open(path, "w"), indent=1)
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
