#!/bin/bash
# kernel trace of the million-triangle render: trace_c5.sh TAG W H S
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace5_$1
mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 -u $GRAFT_REPO_ROOT/tools/gpu_c5.py $2 $3 $4 > $OUT/log.txt 2>&1
grep "render:" $OUT/log.txt | cut -c100-
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
head -6 $f | cut -c1-160
