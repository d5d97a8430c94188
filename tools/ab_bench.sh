#!/bin/bash
# A/B of library variants on bench.py: ab_bench.sh "<bench args>" NAME...   ("base" = the in-tree build)
ARGS=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset RRT_HIP_LIBRARY; else export RRT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_variants/librrt_$v.so; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %8.0f Msamples/s %8.0f Mrays/s %8.2f ms/step frac %.3f' % ('$v', d['value'], d['mrays_per_s'], d['ms_per_step'], d['roofline']['frac']))"
done
