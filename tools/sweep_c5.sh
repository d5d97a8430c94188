#!/bin/bash
# parameter sweep of the tree traversal kernel on the million-triangle scene
run() { timeout -k 10 100 python3 -u tools/gpu_c5.py 1920 1080 8 2>&1 | grep "render:" | tail -1 | cut -c100-; }
export RRT_DEBUG_WF_CAPACITY=16777216
for rm in 1 6 12 24 40; do echo "refill_min=$rm"; RRT_DEBUG_REFILL_MIN=$rm run; done
for lm in 1 2 8; do echo "leaf_max=$lm"; RRT_DEBUG_LEAF_MAX=$lm run; done
for cap in 4194304 33554432 67108864; do echo "capacity=$cap"; RRT_DEBUG_WF_CAPACITY=$cap run; done
