export TMPDIR=/tmp
for s in 16 32 128; do
  rm -rf gpurun_out/clk_$s
  timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/clk_$s -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --spp-side $s > gpurun_out/clk_$s.log 2>&1 || { echo fail; tail -3 gpurun_out/clk_$s.log; exit 1; }
  python3 - $s <<'PY'
import csv, glob, sys
s = sys.argv[1]
cnt = {}
for f in glob.glob(f"gpurun_out/clk_{s}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_fused" in r["Kernel_Name"]:
            cnt.setdefault(r["Dispatch_Id"], 0.0)
            cnt[r["Dispatch_Id"]] += float(r["Counter_Value"])
dur = {}
for f in glob.glob(f"gpurun_out/clk_{s}/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_fused" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for k in sorted(cnt, key=int)[-3:]:
    if k in dur:
        print("side", s, "dispatch", k, "kernel %.3f ms, GRBM_GUI_ACTIVE / 8 / time = %.3f GHz" % (dur[k] * 1e3, cnt[k] / 8 / dur[k] / 1e9))
PY
done
