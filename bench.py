#!/usr/bin/env python3
"""Headline benchmark: Cornell Box 500x500, stratified 16384 spp per GPU (MCPT, depth 2/6).

    python bench.py --gpus N --steps K --warmup W

One step = one full pass of the hot path: every rank renders its 16384 sample
indices of every pixel into an HBM-resident accumulator (rrt_render_accumulate),
the accumulators are summed on rank 0 (RCCL reduce, N > 1) and rank 0 runs the
fused post-process.  Weak scaling: per-GPU work is fixed, the image gets
N x 16384 samples per pixel.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32 peak (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)

# Algorithmic FLOPs, counted from the reference's arithmetic as written
# (1 per add/sub/mul/div/sqrt, FMA = 2, dead code excluded).
FLOPS_PER_PRIM_TEST = {"sphere": 21, "plane": 14, "triangle": 51, "quad": 52}
FLOPS_PER_BOUNCE = [0, 110, 140, 30, 90]  # emissive, lambertian, phong, reflector, dielectric
FLOPS_RAYGEN = 25


def scene_flops_per_ray(flat):
    import numpy as np

    names = ["sphere", "plane", "triangle", "quad"]
    counts = np.bincount(flat.prims["type"], minlength=4)
    return int(sum(FLOPS_PER_PRIM_TEST[names[k]] * int(counts[k]) for k in range(4)))


def algorithmic_flops(stats, flops_per_ray):
    return (stats["rays"] * flops_per_ray + sum(b * f for b, f in zip(stats["bounces"], FLOPS_PER_BOUNCE))
            + stats["samples"] * FLOPS_RAYGEN)


def cpu_baseline(flat, threads):
    """The f64 CPU restatement of the reference renderer (oracle/), timed on a bounded sample."""
    from oracle import oracle_py as O
    from rustraytracer_amd import config

    rc = config.stratified_config(500, 500, 32, 32)  # same scene / image / tracer, 1024 spp
    t0 = time.perf_counter()
    _, st = O.render_accumulate(flat, rc.render_desc(), n_threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": st["samples"] / dt / 1e6,
        "unit": "Msamples/s",
        "mrays_per_s": st["rays"] / dt / 1e6,
        "cores": threads,
        "kind": "port",
        "sample": "Cornell Box 500x500 stratified 32x32 = 1024 spp (256 M samples), f64 C restatement of the "
                  "reference renderer, 16x16 tiles over %d threads, %.1f s" % (threads, dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-side", type=int, default=128, help="strata per axis per GPU (128 -> 16384 spp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    from rustraytracer_amd import api, config, distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP backend has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    flat = config.load_scene(os.path.join(ROOT, "configs", "cornell_box", "scene.toml"))
    be = api.Backend(local_rank)
    scene = api.Scene(flat)
    stream = torch.cuda.current_stream()
    be.set_stream(stream.cuda_stream)

    side = args.spp_side
    xs, ys = D.weak_scaling_strata(side, side, world)
    rc = config.stratified_config(500, 500, xs, ys)
    lo, hi = D.shard_range(rc.spp, rank, world)
    desc = rc.render_desc(sample_begin=lo, sample_end=hi)
    post = rc.post_desc()
    acc = torch.zeros((rc.height, rc.width, 3), dtype=torch.float32, device="cuda")
    img = torch.zeros((rc.height, rc.width, 3), dtype=torch.uint8, device="cuda")

    def step():
        acc.zero_()
        _, st = be.render_accumulate(scene, desc, acc)
        D.reduce_sum(acc, dst=0)
        if rank == 0:
            be.postprocess(acc, rc.spp, post, out=img)
        return st

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    tot = {"samples": 0, "rays": 0, "bounces": [0] * 5, "kernel_ms": 0.0, "nonfinite": 0}
    for _ in range(args.steps):
        st = step()
        tot["samples"] += st["samples"]
        tot["rays"] += st["rays"]
        tot["kernel_ms"] += st["kernel_ms"]
        tot["nonfinite"] += st["nonfinite"]
        tot["bounces"] = [a + b for a, b in zip(tot["bounces"], st["bounces"])]
    fence()
    elapsed = time.perf_counter() - t0

    counts = torch.tensor([tot["samples"], tot["rays"]], dtype=torch.float64, device="cuda")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    all_samples, all_rays = float(counts[0].item()), float(counts[1].item())

    if rank == 0:
        fpr = scene_flops_per_ray(flat)
        flops = algorithmic_flops(tot, fpr)  # rank 0's launches
        kernel_s = tot["kernel_ms"] / 1e3
        achieved = flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s (whole node), Cornell Box 500x500 MCPT",
            "value": all_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "mrays_per_s": all_rays / elapsed / 1e6,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (Cornell Box scene of the reference configs; PLY meshes reconstructed from the "
                    "canonical measurements because the reference ships LFS stubs)",
            "config": {
                "workload": "configs/cornell_box scene, 500x500, stratified %dx%d = %d spp in total, %d sample "
                            "indices per GPU, MCPT min_depth 2 / max_depth 6" % (xs, ys, rc.spp, hi - lo),
                "pipeline": "fused (register-resident paths, lane regeneration)",
                "reduce": "RCCL sum-reduce of %.1f MB fp32 accumulators to rank 0" % (D.message_bytes(500, 500) / 1e6)
                if world > 1 else "none",
                "samples_per_step": int(all_samples / args.steps),
                "rays_per_sample": all_rays / all_samples,
                "nonfinite_samples": tot["nonfinite"],
            },
            "roofline": {
                "bound": "valu_fp32",
                "kernel": "render_fused<float>",
                "achieved": achieved,
                "peak": PEAK_FP32_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                "traffic": traffic,
                "algorithmic_flops_per_launch": flops / args.steps,
                "flops_per_ray_scene": fpr,
                "avg_launch_ms": tot["kernel_ms"] / args.steps,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            out["cpu_baseline"] = cpu_baseline(flat, threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
