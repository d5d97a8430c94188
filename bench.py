#!/usr/bin/env python3
"""Headline benchmark: Cornell Box 500x500 at 16384 spp (stratified 128x128), MCPT depth 2/6, on N GPUs.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak] [--workload NAME]
                    [--pipeline auto|fused|wavefront] [--precision f32|f64]

One step = one full pass of the hot path over one image: every rank renders its sample indices of
every pixel into an HBM-resident accumulator (rrt_render_accumulate), the accumulators are summed on
rank 0 (one RCCL reduce, N > 1) and rank 0 runs the post-process.  Rank 0 prints ONE JSON line.

--scaling strong (default): ONE image at the workload's samples per pixel (16384 for the headline), its
    sample indices split over the ranks (distributed.shard_range) -- one image, one spp, as the
    reference renders it (src/renderer.rs:101-128 splits the same image into tiles over its threads).
--scaling weak: every rank traces the workload's full sample count; the image gets N times as many.

Started without a launcher (`python bench.py --gpus 8`), the script starts its N ranks itself as fresh
child processes through torch.distributed.run, before it touches a GPU, relays rank 0's line and exits
with the launcher's status.  Started by a launcher (WORLD_SIZE set), it is one of the ranks.

Workloads (the default is the one the headline metric is quoted on):
    cornell       configs/cornell_box, 500x500, 128x128 strata
    smallpt       configs/smallpt, 1024x768, 128x128 strata
    cornell_cube  configs/cornell_cube (Cornell box + assets/test.ply), 1024x1024, 64x64 strata
    mesh_1m       configs/mesh_1m (Cornell shell + two generated 500k-triangle meshes), 3840x2160,
                  32x32 strata; closest hit through the surface-area tree
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32 peak (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz, packed FP32)
PEAK_FP64_VALU_TFLOPS = 78.6   # vector FP64: one FMA per lane and clock
PEAK_HBM_GBPS = 8000.0         # HBM3E

# Algorithmic FLOPs, counted from the reference's arithmetic as written
# (1 per add/sub/mul/div/sqrt, FMA = 2, dead code excluded).
FLOPS_PER_PRIM_TEST = {"sphere": 21, "plane": 14, "triangle": 51, "quad": 52}
FLOPS_PER_BOUNCE = [0, 110, 140, 30, 90]  # emissive, lambertian, phong, reflector, dielectric
FLOPS_RAYGEN = 25
# What the kernels execute per primitive test (plane-form tests, rrt_device.hpp): counted from the source the same
# way, comparisons and selects excluded.  sphere: 3 sub + 5 dot + 6 perp + 5 dot + 1 sub + sqrt + 2; plane: 5 + 6 + rcp
# + mul; triangle / parallelogram: 5 (n.d) + 6 (n.o) + rcp + mul + 6 (P) + 6 + 6 (two coefficients) + 3-5 (edge terms).
EXECUTED_FLOPS_PER_PRIM_TEST = {"sphere": 24, "plane": 13, "triangle": 36, "quad": 36}

FLOPS_PER_NODE_TEST = 24  # one ray/box slab test
QUEUE_BYTES_PER_RAY = 88  # wavefront pipeline: ray 24 B + throughput 12 B + ids 8 B, written and read once per bounce

# name -> (scene file, width, height, strata per axis, CPU-baseline strata per axis or (w, h, side))
WORKLOADS = {
    "cornell": ("configs/cornell_box/scene.toml", 500, 500, 128, 32),
    "smallpt": ("configs/smallpt/scene.toml", 1024, 768, 128, 16),
    "cornell_cube": ("configs/cornell_cube/scene.toml", 1024, 1024, 64, 16),
    # the reference scans 1 M triangles per ray (~10 ms each on a host core): its sample is 64x36 pixels at 4 spp
    "mesh_1m": ("configs/mesh_1m/scene.toml", 3840, 2160, 32, (64, 36, 2)),
}
METRIC_NAMES = {"cornell": "Cornell Box 500x500", "smallpt": "smallpt 1024x768",
                "cornell_cube": "Cornell Box + test.ply 1024x1024", "mesh_1m": "1M-triangle mesh scene 3840x2160"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: one image at the workload's spp, sample indices split over the GPUs; "
                         "weak: the workload's spp per GPU")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cornell")
    ap.add_argument("--pipeline", choices=["auto", "fused", "wavefront"], default="auto")
    ap.add_argument("--precision", choices=["f32", "f64"], default="f32",
                    help="f32 is the product path; f64 runs the same kernels in the reference's arithmetic type")
    ap.add_argument("--spp-side", type=int, default=0, help="strata per axis (default: the workload's)")
    ap.add_argument("--exact-hits", action="store_true",
                    help="RRT_RENDER_EXACT_HITS: every f32 closest-hit query tracked and repeated in f64 where f32 cannot decide "
                         "(a measurement of that mode's cost; the headline number is quoted without it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------- launcher (parent process)
def launch_ranks(n, argv):
    """Runs the N ranks of this script as fresh child processes (torch.distributed.run, one per GPU) and relays rank 0's
    JSON line.  The parent never imports torch and never touches a GPU; the children start from a clean process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC, which RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if proc.returncode != 0:
        print("bench.py: a rank failed (launcher exit status %d)" % proc.returncode, file=sys.stderr)
        return proc.returncode
    if not lines:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


# ---------------------------------------------------------------------------- roofline bookkeeping
def scene_flops(flat, table=FLOPS_PER_PRIM_TEST):
    """(flops of a linear scan of the whole scene, mean flops of one primitive test)."""
    import numpy as np

    names = ["sphere", "plane", "triangle", "quad"]
    counts = np.bincount(flat.prims["type"], minlength=4)
    scan = int(sum(table[names[k]] * int(counts[k]) for k in range(4)))
    return scan, scan / max(int(counts.sum()), 1)


def algorithmic_flops(stats, flat, table=FLOPS_PER_PRIM_TEST):
    """Closest-hit arithmetic actually asked for (every primitive per ray in a linear scan; node and
    primitive tests from the device counters through the tree) + shading + ray generation."""
    scan, per_prim = scene_flops(flat, table)
    if stats["node_tests"]:
        hit = stats["node_tests"] * FLOPS_PER_NODE_TEST + stats["prim_tests"] * per_prim
    else:
        hit = stats["rays"] * scan
    return hit + sum(b * f for b, f in zip(stats["bounces"], FLOPS_PER_BOUNCE)) + stats["samples"] * FLOPS_RAYGEN


def measured_traffic(workload, wavefront, precision, rays_per_launch):
    """HBM-side bytes per launch: the bytes per ray of the kept PMC passes of this workload (profiles/hbm_traffic.json,
    written by tools/summarize_profile.py from separate FETCH_SIZE / WRITE_SIZE runs) times the rays of this run's
    launch.  A kept figure of an earlier profile of the same kernels, not a counter of this run.
    -> (bytes or None, source)."""
    key = workload + ("_wavefront" if wavefront and workload != "mesh_1m" else "") + ("_f64" if precision != "f32" else "")
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            table = json.load(f)
    except Exception as e:  # noqa: BLE001
        return None, "profiles/hbm_traffic.json unreadable: %s" % e
    ent = table.get("workloads", {}).get(key)
    if ent is None:
        return None, "no PMC pass kept for %s" % key
    return ent["bytes_per_ray"] * rays_per_launch, ("kept PMC figure, not measured in this run: %.1f B per ray (%s) x the rays of "
                                                    "one step" % (ent["bytes_per_ray"], ent.get("source")))


# ---------------------------------------------------------------------------- CPU baseline
def usable_cpus():
    """Logical CPUs this job may use: the affinity mask, cut to the cgroup's CPU quota and to the share the box
    announces through OMP_NUM_THREADS (on a shared GPU host os.cpu_count() reports the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(n, 1)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:  # noqa: BLE001
        pass
    return "unknown"


def cpu_baseline(flat, name, width, height, side, threads):
    """The f64 CPU restatement of the reference renderer (oracle/), timed on a bounded sample, with
    `parallelism = 0` semantics: every CPU the job may use (src/renderer.rs:74-78)."""
    from oracle import oracle_py as O
    from rustraytracer_amd import config

    rc = config.stratified_config(width, height, side, side)  # same scene / image / tracer, fewer spp
    t0 = time.perf_counter()
    _, st = O.render_accumulate(flat, rc.render_desc(), n_threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": st["samples"] / dt / 1e6,
        "unit": "Msamples/s",
        "mrays_per_s": st["rays"] / dt / 1e6,
        "cores": threads,
        "host_logical_cpus": os.cpu_count(),
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": "%s %dx%d stratified %dx%d = %d spp (%.4g M samples), f64 C restatement of the reference "
                  "renderer (linear scan of every primitive per ray), 16x16 tiles over %d threads = every CPU "
                  "this job may use (the host has %d logical CPUs), %.1f s"
                  % (name, width, height, side, side, side * side, st["samples"] / 1e6, threads, os.cpu_count() or 0, dt),
    }


# ---------------------------------------------------------------------------- one rank
def run_rank(args):
    import numpy as np  # noqa: F401
    import torch

    from rustraytracer_amd import config, distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # RRT_BENCH_REHEARSAL=1: every rank on GPU 0, gloo instead of RCCL -- the multi-rank code path (shards, reduce,
    # max-over-ranks timing) on a one-GPU box.  RRT_BENCH_REHEARSAL=plumbing: no GPU at all -- launch, process group,
    # shard plan and reduce only, with each rank's accumulator filled with its sample count instead of rendered.
    # Either way the line is marked and is no result.
    rehearsal = os.environ.get("RRT_BENCH_REHEARSAL", "")
    plumbing = rehearsal == "plumbing"
    if not plumbing and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP backend has no CPU fallback)")
    if rehearsal:
        local_rank = 0
    dev = "cpu" if plumbing else "cuda"
    if not plumbing:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    scene_file, width, height, default_side, cpu_side = WORKLOADS[args.workload]
    if args.workload == "mesh_1m":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_c5_assets

        if local_rank == 0 and (rank == 0 or not rehearsal):
            make_c5_assets.main()  # deterministic generator; the 25 MB of PLY are not committed
        if world > 1:
            dist.barrier()
    flat = config.load_scene(os.path.join(ROOT, scene_file))

    side = args.spp_side or default_side
    if args.scaling == "weak":
        xs, ys = D.weak_scaling_strata(side, side, world)
    else:
        xs, ys = side, side
    rc = config.stratified_config(width, height, xs, ys)
    lo, hi = D.shard_range(rc.spp, rank, world)
    if hi <= lo:
        raise SystemExit("rank %d of %d has no sample index of %d spp" % (rank, world, rc.spp))

    reduce_ms = [0.0]  # time of the sum-reduce, summed over the timed steps (this rank's clock)
    if plumbing:
        acc = torch.zeros((rc.height, rc.width, 3), dtype=torch.float32)

        def step():
            acc.fill_(float(hi - lo))
            t_r = time.perf_counter()
            D.reduce_sum(acc, dst=0)
            reduce_ms[0] += (time.perf_counter() - t_r) * 1e3
            if rank == 0 and not bool((acc == float(rc.spp)).all()):
                raise SystemExit("the shards do not add up to %d samples per pixel" % rc.spp)
            n = (hi - lo) * rc.width * rc.height
            return {"samples": n, "rays": n, "kernel_ms": 0.0, "nonfinite": 0, "node_tests": 0, "prim_tests": 0,
                    "bounces": [0] * 5, "accel_used": 1, "pipeline_used": 1, "tree_depth": 0, "tree_built": 0,
                    "tree_build_ms": 0.0, "stack_overflows": 0, "exact_redos": 0}
    else:
        from rustraytracer_amd import api

        be = api.Backend(local_rank)
        scene = api.Scene(flat)
        be.set_stream(torch.cuda.current_stream().cuda_stream)
        pipeline = {"auto": api.PIPELINE_AUTO, "fused": api.PIPELINE_FUSED, "wavefront": api.PIPELINE_WAVEFRONT}[args.pipeline]
        precision = api.PRECISION_F64 if args.precision == "f64" else api.PRECISION_F32
        desc = rc.render_desc(sample_begin=lo, sample_end=hi, pipeline=pipeline, precision=precision,
                              flags=api.RENDER_EXACT_HITS if args.exact_hits else 0)
        post = rc.post_desc()
        acc = torch.zeros((rc.height, rc.width, 3), dtype=torch.float32, device="cuda")
        img = torch.zeros((rc.height, rc.width, 3), dtype=torch.uint8, device="cuda")

        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def step():
            acc.zero_()
            _, st = be.render_accumulate(scene, desc, acc)
            if world > 1:
                # the library's kernels and the reduce run on torch's current stream: events on it bracket the reduce
                ev0.record()
                if rehearsal:
                    host = acc.cpu()
                    D.reduce_sum(host, dst=0)
                    acc.copy_(host)
                else:
                    D.reduce_sum(acc, dst=0)
                ev1.record()
                ev1.synchronize()
                reduce_ms[0] += ev0.elapsed_time(ev1)
            if rank == 0:
                be.postprocess(acc, rc.spp, post, out=img)
            return st

    def fence():
        if dist is not None:
            dist.barrier()
        if not plumbing:
            torch.cuda.synchronize()

    first = None  # the statistics of the first pass: which tree it built and how long that took
    for _ in range(args.warmup):
        st = step()
        first = first or st
    fence()
    reduce_ms[0] = 0.0
    t0 = time.perf_counter()
    tot = {"samples": 0, "rays": 0, "bounces": [0] * 5, "kernel_ms": 0.0, "nonfinite": 0, "node_tests": 0,
           "prim_tests": 0}
    for _ in range(args.steps):
        st = step()
        for k in ("samples", "rays", "kernel_ms", "nonfinite", "node_tests", "prim_tests"):
            tot[k] += st[k]
        tot["bounces"] = [a + b for a, b in zip(tot["bounces"], st["bounces"])]
        first = first or st
    fence()
    elapsed = time.perf_counter() - t0
    # What ran, as the library reports it (rrt_stats), not as this script would guess it from the scene size.
    ACCEL_NAMES = {1: "linear", 2: "lbvh", 3: "sah", 4: "sah_gpu"}
    accel_used = ACCEL_NAMES.get(st["accel_used"], "?")
    use_tree = accel_used != "linear"
    wavefront = st["pipeline_used"] == 2

    cdev = "cpu" if rehearsal else dev
    counts = torch.tensor([tot["samples"], tot["rays"], 1.0], dtype=torch.float64, device=cdev)
    tmax = torch.tensor([elapsed, tot["kernel_ms"]], dtype=torch.float64, device=cdev)
    if dist is not None:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax[0].item())
    all_samples, all_rays, ranks_seen = float(counts[0].item()), float(counts[1].item()), int(round(float(counts[2].item())))
    per_rank = None
    if dist is not None:
        # every rank's own kernel time and reduce time per step, so that a scaling curve explains itself: a slow rank,
        # a slow reduce and launch overhead look the same in the max-over-ranks wall time
        mine = torch.tensor([tot["kernel_ms"] / max(args.steps, 1), reduce_ms[0] / max(args.steps, 1)], dtype=torch.float64, device=cdev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = [[float(g[0].item()), float(g[1].item())] for g in gathered]

    if rank == 0:
        fpr = scene_flops(flat)[0]
        flops = algorithmic_flops(tot, flat)  # rank 0's launches
        executed = algorithmic_flops(tot, flat, EXECUTED_FLOPS_PER_PRIM_TEST)
        kernel_s = tot["kernel_ms"] / 1e3
        achieved = flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
        peak = PEAK_FP32_VALU_TFLOPS if args.precision == "f32" else PEAK_FP64_VALU_TFLOPS
        traffic, traffic_source = measured_traffic(args.workload, wavefront, args.precision, tot["rays"] / max(args.steps, 1))
        out = {
            "metric": "Msamples/s (whole node), %s MCPT" % METRIC_NAMES[args.workload],
            "value": all_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "mrays_per_s": all_rays / elapsed / 1e6,
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (scenes of the reference configs; Cornell PLY meshes reconstructed from the canonical "
                    "measurements because the reference ships LFS stubs; the 1M-triangle meshes are generated)",
            "config": {
                "workload": "%s, %dx%d, stratified %dx%d = %d spp in total, sample indices [%d, %d) of every pixel on "
                            "rank 0 (%d per GPU), MCPT min_depth 2 / max_depth 6"
                            % (scene_file, width, height, xs, ys, rc.spp, lo, hi, hi - lo),
                "pipeline": "wavefront (SoA ray/hit queues in HBM, ballot/rank compaction)" if wavefront
                else "fused (register-resident paths, lane regeneration)",
                "closest_hit": {"linear": "linear scan (wave-uniform)",
                                "lbvh": "tree built on the GPU: Morton order + bottom-up clustering, 4-wide quantised nodes",
                                "sah": "tree built by the HOST threads (binned surface-area heuristic), 4-wide quantised nodes widened and packed on the GPU",
                                "sah_gpu": "tree built on the GPU: binned surface-area heuristic level by level, small ranges by an exact sweep, 4-wide quantised nodes",
                                }.get(accel_used, accel_used),
                "accel_used": accel_used,
                "pipeline_used": "wavefront" if wavefront else "fused",
                "tree_depth": st["tree_depth"],
                "tree_build_ms_first_call": first["tree_build_ms"] if first and first["tree_built"] else None,
                "stack_overflows": st["stack_overflows"],
                "exact_hits": bool(args.exact_hits),
                "exact_redos_per_ray": st["exact_redos"] / max(st["rays"], 1),
                "reduce": "RCCL sum-reduce of %.1f MB fp32 accumulators to rank 0" % (D.message_bytes(width, height) / 1e6)
                if world > 1 else "none",
                "samples_per_step": int(all_samples / args.steps),
                "rays_per_sample": all_rays / max(all_samples, 1),
                "nonfinite_samples": tot["nonfinite"],
            },
            "roofline": {
                "bound": "valu_fp32" if args.precision == "f32" else "valu_fp64",
                "kernel": ("wf_closest_hit_tree + wf_shade" if use_tree else "wf_shade<scan>") if wavefront
                else "render_fused<%s>" % ("float" if args.precision == "f32" else "double"),
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "flops_counted_as": "the reference's arithmetic per primitive test (Moller-Trumbore 51, Lagae-Dutre 52, sphere 21, "
                                    "plane 14), shading and ray generation: algorithmic work over peak, not hardware busy time",
                "algorithmic_flops_per_launch": flops / args.steps,
                "executed_flops_per_launch": executed / args.steps,
                "executed_tflops": executed / kernel_s / 1e12 if kernel_s > 0 else 0.0,
                "executed_note": "the kernels' own plane-form tests need about two thirds of the reference's operations per "
                                 "primitive; executed_tflops / peak is the share of FP32 issue slots doing that arithmetic",
                "flops_per_ray_scene": fpr,
                "avg_launch_ms": tot["kernel_ms"] / args.steps,
            },
        }
        if per_rank is not None:
            out["per_rank_kernel_ms"] = [p[0] for p in per_rank]
            out["per_rank_reduce_ms"] = [p[1] for p in per_rank]
            out["reduce_ms"] = max(p[1] for p in per_rank)
        if rehearsal:
            out["rehearsal"] = ("plumbing only: no GPU, nothing rendered, not a measurement" if plumbing
                                else "all ranks on GPU 0, gloo: not a measurement")
        if use_tree:
            out["config"]["node_tests_per_ray"] = tot["node_tests"] / max(tot["rays"], 1)
            out["config"]["prim_tests_per_ray"] = tot["prim_tests"] / max(tot["rays"], 1)
            # every traversal step fetches one 64-byte record (wide node or leaf primitive) per lane: the tree
            # kernel is bound by how fast the memory system serves such divergent gathers (L2 + HBM), not by flops
            steps = tot["node_tests"] + tot["prim_tests"]
            if kernel_s > 0:
                out["roofline"]["tree_records_per_s"] = steps / kernel_s
                out["roofline"]["tree_record_traffic_GBps"] = steps * 64 / kernel_s / 1e9
        if wavefront and kernel_s > 0:
            out["roofline"]["queue_traffic_algorithmic_GBps"] = tot["rays"] * QUEUE_BYTES_PER_RAY / kernel_s / 1e9
        if traffic and kernel_s > 0:
            # the competing bound: HBM-side bytes (kept PMC figure per ray) over this run's kernel time against ~8 TB/s
            out["roofline"]["hbm_GBps"] = traffic * args.steps / kernel_s / 1e9
            out["roofline"]["hbm_frac"] = out["roofline"]["hbm_GBps"] / PEAK_HBM_GBPS
        if world == 1 and not args.no_cpu_baseline and not plumbing:
            cw, ch, cs = cpu_side if isinstance(cpu_side, tuple) else (width, height, cpu_side)
            out["cpu_baseline"] = cpu_baseline(flat, args.workload, cw, ch, cs, usable_cpus())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # before any import of torch, before any GPU call
    run_rank(args)


if __name__ == "__main__":
    main()
