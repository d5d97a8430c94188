#!/bin/bash
# A/B of library variants on the million-triangle scene: ab_c5.sh NAME...  ("base" = the in-tree build)
for v in "$@"; do
  echo "variant=$v"
  if [ "$v" = base ]; then unset RRT_HIP_LIBRARY; else export RRT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_variants/librrt_$v.so; fi
  timeout -k 10 100 python3 -u tools/gpu_c5.py 1920 1080 8 2>&1 | grep "render:" | tail -1 | cut -c100-
done
