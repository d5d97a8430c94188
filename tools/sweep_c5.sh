#!/bin/bash
# parameter sweep of the tree traversal kernel on the million-triangle scene
for lm in 1 2 4 8; do echo "leaf_max=$lm"; RRT_DEBUG_LEAF_MAX=$lm timeout -k 10 100 python3 -u tools/gpu_c5.py 960 540 8 2>&1 | grep "render:" | tail -1 | cut -c100-; done
for cap in 1048576 16777216; do echo "capacity=$cap"; RRT_DEBUG_WF_CAPACITY=$cap timeout -k 10 100 python3 -u tools/gpu_c5.py 960 540 8 2>&1 | grep "render:" | tail -1 | cut -c150-; done
echo "big image 16M"; RRT_DEBUG_WF_CAPACITY=16777216 timeout -k 10 100 python3 -u tools/gpu_c5.py 1920 1080 8 2>&1 | grep "render:" | tail -1 | cut -c100-
