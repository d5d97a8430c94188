#!/bin/bash
# A/B of library variants on the million-triangle scene (1920x1080 at 8x8 strata; -f: 3840x2160 at 32x32):
#   ab_lib_env.sh [-f] NAME...      ("base" = the in-tree build)
W=1920; H=1080; S=8
if [ "$1" = "-f" ]; then W=3840; H=2160; S=32; shift; fi
for v in "$@"; do
  if [ "$v" = base ]; then unset RRT_HIP_LIBRARY; else export RRT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_variants/librrt_$v.so; fi
  echo "library $v"
  timeout -k 10 250 python3 -u tools/gpu_c5.py $W $H $S 2>&1 | grep "render:" | tail -1 | cut -c100- || exit 1
done
