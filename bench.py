#!/usr/bin/env python3
"""Throughput bench of the path-tracing core.

    python bench.py --gpus N --steps K --warmup W [--workload mesh|cornell|teaset|closest_hit] ...

One step = one render pass (`--spp` samples per pixel over the whole frame, `--bounces`
bounces) of the workload's scene; passes are sharded round-robin over the ranks
(rank r renders passes r, r+N, ...), each rank holds a replica of the scene, and the
float accumulators are summed once with a RCCL reduce at the end of the timed region.
Prints ONE JSON line on rank 0:

  value        Mrays/s of the whole job: Bvh::hit invocations (primary rays + every continued
               bounce) of all ranks / max-over-ranks wall time of the K steps
  roofline     closest-hit ("extend") kernel: algorithmic bytes per ray (node visits x 128 B +
               primitive tests x 48 B + 32 B ray in + 8 B hit out, visits/tests counted by the
               instrumented kernel on the same workload outside the timed region) x rays per
               launch / mean launch duration (HIP events on the launch stream, timed region)
  cpu_baseline the CPU oracle (a restatement of the reference's renderer, not the Rust binary)
               on a bounded sample of the same workload: same scene, samples and bounces at a
               reduced resolution, on this box's host cores (N=1 only)
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
LIMITER = None  # what the PMC profile says bounds the kernel, when one is committed for the configuration


def build_workload(pkg, args):
    if args.workload == "mesh" or args.workload == "closest_hit":
        desc = pkg.scenes.mesh_scene(n_triangles=args.tris, seed=7)
        name = "synthetic %d-triangle mesh scene (displaced spheres in a lit room, mixed materials)" % desc.num_primitives()
    elif args.workload == "cornell":
        desc = pkg.scenes.cornell()
        name = "Cornell box (values of scene/example/cornell.ron)"
    elif args.workload == "teaset":
        desc = pkg.scenes.teaset()
        name = "Newell tea set (8 OBJ models, 17,228 triangles, glass + ceramic + steel) under a procedural HDR sky"
    elif args.workload == "rtiow":
        desc = pkg.scenes.rtiow(seed=1)
        name = "procedural RTIOW spheres scene (seeded)"
    else:
        raise SystemExit("unknown workload " + args.workload)
    return desc, name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mesh")
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--rays", type=int, default=100_000_000, help="closest_hit workload: rays per step")
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the core has no CPU path")
    # one process per GPU; RT_BENCH_BACKEND=gloo lets several ranks share one card to rehearse the N > 1 control flow
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge

    pkg = ge.load_package()
    lib = pkg.load()
    pkg._abi.check(lib.rt_init(device_index))

    t0 = time.time()
    desc, wl_name = build_workload(pkg, args)
    t_gen = time.time() - t0
    t0 = time.time()
    scene = pkg.Scene(desc)
    t_build = time.time() - t0
    info = scene.info()
    stream = torch.cuda.current_stream().cuda_stream

    W, H, spp, bounces = args.width, args.height, args.spp, args.bounces

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "closest_hit":
        n_rays = args.rays
        mn, mx = np.array(info.bounds_min[:]), np.array(info.bounds_max[:])
        g = torch.Generator(device="cuda")
        g.manual_seed(0x5EED + rank)
        c, h = torch.tensor((mn + mx) / 2, device="cuda"), torch.tensor((mx - mn) / 2 * 1.5, device="cuda")
        org = c + (torch.rand((n_rays, 3), device="cuda", generator=g) * 2 - 1) * h
        z = torch.rand(n_rays, device="cuda", generator=g) * 2 - 1
        ph = torch.rand(n_rays, device="cuda", generator=g) * (2 * math.pi)
        rr = torch.sqrt(torch.clamp(1 - z * z, min=0))
        rays = torch.cat([org, torch.stack([rr * torch.cos(ph), rr * torch.sin(ph), z], dim=1)], dim=1).float().contiguous()
        ids = torch.empty(n_rays, dtype=torch.int32, device="cuda")
        tt = torch.empty(n_rays, dtype=torch.float32, device="cuda")
        nrm = torch.empty((n_rays, 3), dtype=torch.float32, device="cuda")

        def step(i, counted=False):
            scene.closest_hit_device(rays.data_ptr(), n_rays, ids.data_ptr(), tt.data_ptr(), nrm.data_ptr(), None, stream,
                                     counted=counted)

        def count_step():
            step(0, counted=True)
    else:
        accum = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")

        def step(i):
            p = pkg.Scene.params(W, H, spp, bounces, pass_index=rank + i * world, seed=1234, batch_paths=args.batch_paths)
            scene.render_pass_device(accum.data_ptr(), p, stream)

        def count_step():
            # same workload at a quarter of the resolution: the same ray distribution, fewer rays
            p = pkg.Scene.params(max(2, W // 2), max(2, H // 2), spp, bounces, pass_index=10_000, seed=1234)
            tmp = torch.zeros((max(2, W // 2) * max(2, H // 2) * 3), dtype=torch.float32, device="cuda")
            scene.set_counting(True)
            scene.render_pass_device(tmp.data_ptr(), p, stream)
            torch.cuda.synchronize()
            scene.set_counting(False)

    # ---- bytes per ray, from the instrumented kernel (untimed)
    scene.stats()
    count_step()
    cs = scene.stats()
    visits_per_ray = cs.node_visits / max(1, cs.rays)
    tests_per_ray = cs.prim_tests / max(1, cs.rays)
    bytes_per_ray = visits_per_ray * info.node_bytes + tests_per_ray * info.primitive_bytes + 32 + 8

    # ---- warmup, then the timed region
    for i in range(args.warmup):
        step(i)
    if world > 1 and args.workload != "closest_hit" and args.warmup > 0:
        # the warmup also runs the collective once, on a scratch buffer of the accumulator's size, so that
        # the communicator's channels and buffers exist before the timed region
        scratch = torch.zeros_like(accum)
        dist.reduce(scratch, dst=0)
        del scratch
    barrier()
    scene.stats()
    scene.set_stage_timing(True)
    t_start = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if world > 1 and args.workload != "closest_hit":
        dist.reduce(accum, dst=0)  # the one collective of a render: sum of the accumulators
    barrier()
    elapsed = time.perf_counter() - t_start
    st = scene.stats()
    scene.set_stage_timing(False)

    t_all = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    r_all = torch.tensor([float(st.rays)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(r_all, op=dist.ReduceOp.SUM)
    elapsed_max = float(t_all.item())
    rays_total = float(r_all.item())

    if rank == 0:
        value = rays_total / elapsed_max / 1e6
        extend_s = st.extend_ms / 1e3
        achieved = (st.rays * bytes_per_ray / extend_s / 1e9) if extend_s > 0 else 0.0
        launches = max(1, int(st.extend_launches))
        traffic, traffic_note = measured_traffic(args, info)
        out = {
            "metric": "Mrays/s (extend+shade)",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl_name,
                "kind": args.workload,
                "width": W, "height": H, "spp_per_pass": spp, "max_bounce": bounces,
                "primitives": int(info.n_primitives), "wide_nodes": int(info.n_wide_nodes),
                "scene_MB": round(info.device_bytes / 1e6, 1),
                "parallelism": "passes round-robin over %d GPU(s), scene replicated, one RCCL reduce" % world,
                "s_per_4k_frame": round(elapsed_max / args.steps, 4) if (W, H) == (3840, 2160) else None,
                "rays_per_step_per_gpu": int(st.rays / max(1, args.steps)),
                "scene_gen_s": round(t_gen, 2), "scene_build_s": round(t_build, 2),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "extend_kernel (closest hit)",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_note": traffic_note,
                "limiter": LIMITER,
                "launches": launches,
                "mean_launch_ms": round(st.extend_ms / launches, 3),
                "algorithmic_bytes_per_launch": int(st.rays * bytes_per_ray / launches),
                "bytes_per_ray": round(bytes_per_ray, 1),
                "node_visits_per_ray": round(visits_per_ray, 2),
                "prim_tests_per_ray": round(tests_per_ray, 2),
                "extend_ms_total": round(st.extend_ms, 3),
                "shade_ms_total": round(st.shade_ms, 3),
                "other_ms_total": round(st.other_ms, 3),
                "extend_Mrays_per_s": round(st.rays / extend_s / 1e6, 1) if extend_s > 0 else None,
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(pkg, desc, args, W, H, spp, bounces, info)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args, info):
    """HBM bytes per closest-hit launch from the PMC pass of the same command (rocprofv3 --pmc FETCH_SIZE in its
    own run, summarised by tools/extend_pmc_json.py), or None when no profile of this configuration is committed."""
    path = os.path.join(ROOT, "profiles", "r01_extend_pmc_c4_final.json")
    default_cfg = (args.workload == "mesh" and args.tris == 10_000_000 and (args.width, args.height) == (3840, 2160)
                   and args.spp == 16 and args.bounces == 8 and args.batch_paths == 0)
    if not (default_cfg and os.path.exists(path)):
        return None, "no PMC profile committed for this configuration"
    p = json.load(open(path))
    global LIMITER
    LIMITER = ("instruction issue, not bandwidth: VALU busy %.0f %% of SIMD cycles at %.0f %% active lanes, L2 hit rate %.0f %% "
               "(same PMC profile); algorithmic bytes exceed HBM traffic because the upper tree levels are served from cache"
               % (100 * p["valu_busy_fraction"], 100 * p["valu_lane_utilisation"], 100 * p["l2_hit_rate"]))
    return (int(p["hbm_bytes_per_launch_corrected"]),
            "bytes per launch from %s: FETCH_SIZE x 1 KiB x 2 (gfx950 reports half the bytes of 16 B/lane reads; "
            "calibrated on coalesced streams, this kernel gathers one line per lane); raw %d" % (
                os.path.relpath(path, ROOT), int(p["hbm_bytes_per_launch_raw"])))


def cpu_baseline(pkg, desc, args, W, H, spp, bounces, info):
    """The CPU oracle on a bounded sample of the workload (checker code, timed here as the CPU column)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    ora = ob.OracleScene(desc)
    if args.workload == "closest_hit":
        n = 200_000
        rays = pkg.scenes.random_rays(n, info.bounds_min[:], info.bounds_max[:])
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < args.cpu_seconds:
            ora.closest_hit(rays, threads=threads)
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": round(n * reps / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": "%d x %d random rays, same scene" % (reps, n)}
    # calibrate on a tiny frame, then size the sample to about cpu_seconds
    w0, h0 = max(2, W // 32), max(2, H // 32)
    p = pkg.Scene.params(w0, h0, spp, bounces, seed=1234)
    acc = np.zeros(w0 * h0 * 3, np.float32)
    t0 = time.perf_counter()
    ora.render_pass(acc, p, threads=threads)
    dt0 = time.perf_counter() - t0
    scale = math.sqrt(max(1.0, args.cpu_seconds / max(dt0, 1e-3)))
    w1, h1 = min(W, max(2, int(w0 * scale))), min(H, max(2, int(h0 * scale)))
    p = pkg.Scene.params(w1, h1, spp, bounces, seed=1234)
    acc = np.zeros(w1 * h1 * 3, np.float32)
    t0 = time.perf_counter()
    ora.render_pass(acc, p, threads=threads)
    dt = time.perf_counter() - t0
    rays = ora.counters()["rays"]
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "one pass of the same scene, camera, %d spp, %d bounces at %dx%d (%.1f s)" % (spp, bounces, w1, h1, dt)}


if __name__ == "__main__":
    main()
