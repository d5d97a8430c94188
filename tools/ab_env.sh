#!/bin/bash
# A/B of an environment switch on the million-triangle scene: ab_env.sh [-f] VAR VALUE...
# (1920x1080 at 8x8 strata; -f: the benchmark's 3840x2160 at 32x32)
W=1920; H=1080; S=8
if [ "$1" = "-f" ]; then W=3840; H=2160; S=32; shift; fi
VAR=$1; shift
for v in "$@"; do
  echo "$VAR=$v"
  env $VAR=$v timeout -k 10 250 python3 -u tools/gpu_c5.py $W $H $S 2>&1 | grep "render:" | tail -1 | cut -c100- || exit 1
done
