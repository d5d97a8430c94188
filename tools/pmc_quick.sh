#!/bin/bash
# quick SQ counter passes on a short bench run (1024 spp per GPU); usage: pmc_quick.sh <tag>
set -u
OUT=gpurun_out/pmcq_${1:-x}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp-side 32"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq1" -- $BENCH > "$OUT/sq1.log" 2>&1 || { echo fail1; tail -3 "$OUT/sq1.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d "$OUT/sq2" -- $BENCH > "$OUT/sq2.log" 2>&1 || { echo fail2; tail -3 "$OUT/sq2.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/grbm" -- $BENCH > "$OUT/grbm.log" 2>&1 || { echo fail3; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
tot = {}
for d in ("sq1", "sq2", "grbm"):
    for f in glob.glob(f"{out}/{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "render_fused" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            tot[k] = sum(v) / len(v)
for k in sorted(tot):
    print(f"{k:28s} {tot[k]:.5g}")
rays = 500*500*1024*3.11
it = rays/64
print("per wave-iteration: VALU %.0f SALU %.0f SMEM %.1f BRANCH %.1f VMEM_RD %.2f" % (tot["SQ_INSTS_VALU"]/it, tot["SQ_INSTS_SALU"]/it, tot["SQ_INSTS_SMEM"]/it, tot.get("SQ_INSTS_BRANCH",0)/it, tot["SQ_INSTS_VMEM_RD"]/it))
print("lane eff %.3f" % (tot["SQ_THREAD_CYCLES_VALU"]/tot["SQ_ACTIVE_INST_VALU"]/64))
wc = tot["SQ_WAVE_CYCLES"]
print("wave-cycle shares: wait_any %.2f wait_inst %.2f active %.2f" % (tot["SQ_WAIT_ANY"]/wc, tot["SQ_WAIT_INST_ANY"]/wc, tot["SQ_ACTIVE_INST_ANY"]/wc))
PY
grep -h ms_per_step "$OUT"/sq1.log | cut -c1-300
