#!/bin/bash
# A/B of library variants on the million-triangle scene: ab_c5.sh NAME...   (env LM = leaf size)
for v in "$@"; do
  echo "variant=$v"
  RRT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_variants/librrt_$v.so RRT_DEBUG_WF_CAPACITY=16777216 timeout -k 10 100 python3 -u tools/gpu_c5.py 1920 1080 8 2>&1 | grep "render:" | tail -1 | cut -c100-
done
