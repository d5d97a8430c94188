#!/bin/bash
# VALU instruction count and lane efficiency of the fused kernel on a short Cornell run (1024 spp)
set -u
OUT=gpurun_out/pmcf_${1:-x}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp-side 64 ${*:2}"  # pmc_fused.sh TAG [more bench arguments, e.g. --workload smallpt]
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq1" -- $BENCH > "$OUT/sq1.log" 2>&1 || { echo fail1; tail -3 "$OUT/sq1.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
tot = collections.defaultdict(list)
for f in glob.glob(f"{out}/sq1/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_fused" in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in tot.items()}
b = json.loads([l for l in open(f"{out}/sq1.log") if l.startswith("{")][-1])
rays = b["mrays_per_s"] * b["ms_per_step"] * 1e3
print("per ray: VALU wave-instr x64 = %.0f lane slots, SALU %.1f, SMEM %.2f; lane eff %.3f; rays/launch %.3g" % (
    c["SQ_INSTS_VALU"] * 64 / rays, c["SQ_INSTS_SALU"] * 64 / rays, c["SQ_INSTS_SMEM"] * 64 / rays,
    c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64, rays))
PY
