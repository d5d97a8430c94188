#!/bin/bash
# kernel trace of the million-triangle render with a library variant: trace_c5.sh VARIANT W H S
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace5_$1
mkdir -p $OUT
cd /tmp
RRT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_variants/librrt_$1.so RRT_DEBUG_WF_CAPACITY=${CAP:-16777216} timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 -u $GRAFT_REPO_ROOT/tools/gpu_c5.py $2 $3 $4 > $OUT/log.txt 2>&1
grep "render:" $OUT/log.txt | cut -c100-
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-200
