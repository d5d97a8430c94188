#!/bin/bash
# PMC passes over the tree-traversal kernel on the million-triangle scene; usage: pmc_c5.sh <tag> [passes...]
set -u
TAG=${1:-x}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc5_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
RUN="python3 -u $GRAFT_REPO_ROOT/tools/gpu_c5.py 960 540 8"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- $RUN > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
if [ $# -eq 0 ]; then set -- sq1 sq2 tcc tcp fetch grbm; fi
for p in "$@"; do
case $p in
sq1) pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD;;
sq2) pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR;;
tcc) pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum;;
tcp) pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum;;
tcp2) pass tcp2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum;;
tcp3) pass tcp3 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum;;
ta) pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum;;
ta2) pass ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum;;
fetch) pass fetch FETCH_SIZE;;
grbm) pass grbm GRBM_GUI_ACTIVE;;
esac
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
tot = {}
for f in glob.glob(f"{out}/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "wf_closest_hit" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in agg.items():
        tot[k] = v; tot["n_" + k] = n[k]
for k in sorted(tot):
    if not k.startswith("n_"): print(f"{k:34s} {tot[k]:.6g}  ({tot['n_'+k]} dispatches)")
PY
grep -h "render:" "$OUT"/sq1.log | cut -c1-250
