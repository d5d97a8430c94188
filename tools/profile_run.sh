#!/bin/bash
# rocprofv3 evidence for one bench.py configuration (run on the GPU box from the repo root):
#   tools/profile_run.sh TAG [bench.py arguments...]
# One --kernel-trace --stats run, then one run per PMC pass (counters are never combined with traces).
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline $*"
echo "$BENCH" > "$OUT/command.txt"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo "profile done"
