#!/bin/bash
# Per-kernel time of library variants on the million-triangle scene (1920x1080 at 16x16 strata), rocprofv3 kernel trace:
#   ab_kernel_split.sh NAME...      ("base" = the in-tree build)
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset RRT_HIP_LIBRARY; else export RRT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_variants/librrt_$v.so; fi
  echo "library $v"
  rm -rf gpurun_out/split_$v
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/split_$v -- python3 tools/gpu_c5.py 1920 1080 16 > gpurun_out/split_$v.log 2>&1 || { tail -5 gpurun_out/split_$v.log; exit 1; }
  f=$(find gpurun_out/split_$v -name '*kernel_stats.csv' | head -1)
  cut -d, -f1-4 "$f" | grep -E "wf_|Name" | cut -c1-150
done
