#!/bin/bash
# One bench.py line per workload (and the f64 and wavefront variants of the headline one) -> gpurun_out/bench_all.jsonl
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/bench_all.jsonl
: > $OUT
run() { timeout -k 10 400 python3 bench.py "$@" 2>/dev/null | tail -1 >> $OUT || exit 1; }
run
run --workload smallpt --steps 2 --warmup 1
run --workload cornell_cube --steps 3 --warmup 1
run --workload mesh_1m --steps 1 --warmup 1
run --pipeline wavefront --steps 3 --warmup 1 --no-cpu-baseline
run --precision f64 --steps 2 --warmup 1 --no-cpu-baseline
python3 - <<'PY'
import json
for l in open("gpurun_out/bench_all.jsonl"):
    d = json.loads(l)
    print("%-62s %9.0f Msamples/s %9.0f Mrays/s %9.1f ms/step  frac %.3f  cpu %s" % (d["metric"][:40] + " " + d["dtype"] + " " + d["config"]["pipeline"][:9],
          d["value"], d["mrays_per_s"], d["ms_per_step"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value")))
PY
