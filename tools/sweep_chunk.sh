#!/bin/bash
# Samples per work item of the fused kernel: sweep_chunk.sh "<bench args>" CHUNK...
ARGS=$1; shift
for c in "$@"; do echo "chunk=$c"; RRT_DEBUG_CHUNK=$c tools/ab_bench.sh "$ARGS" base; done
