#!/bin/bash
# Queue capacity of the wavefront pipeline on the million-triangle scene: ab_cap.sh W H S CAP...
W=$1; H=$2; S=$3; shift 3
for c in "$@"; do
  echo "capacity=$c"
  RRT_DEBUG_WF_CAPACITY=$c timeout -k 10 200 python3 -u tools/gpu_c5.py $W $H $S 2>&1 | grep "render:" | tail -1 | cut -c100- || exit 1
done
