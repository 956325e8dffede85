#!/usr/bin/env python3
"""Throughput bench of the path-tracing core.

    python bench.py --gpus N --steps K --warmup W [--workload mesh|cornell|teaset|closest_hit] ...

One step = one render pass (`--spp` samples per pixel over the whole frame, `--bounces`
bounces) of the workload's scene; passes are sharded round-robin over the ranks
(rank r renders passes r, r+N, ...), each rank holds a replica of the scene, and the
float accumulators are summed once at the end of the timed region (RCCL).
Prints ONE JSON line on rank 0:

  value        Mrays/s of the whole job: Bvh::hit invocations (primary rays + every continued
               bounce) of all ranks / max-over-ranks wall time of the K steps
  roofline     closest-hit ("extend") kernel, from the rocprofv3 counter profile of this same configuration under
               profiles/, used only while its stamp (hash of the kernel sources, bench arguments) matches this run.  Three
               units can bound it and all are reported; `bound` names the busiest one:
                 "valu"  the vector ALU, against a MEASURED ceiling (profiles/r03_valu_rate.json: the node step's own
                         code on register-resident nodes, in-kernel cycle counter, at the kernel's waves per SIMD):
                         peak = every SIMD issuing vector instructions at the ceiling rate with all 64 lanes useful,
                         achieved = vector instructions x active lanes per instruction,
                         frac = issue_frac (ceiling cycles per instruction / the kernel's) x lane utilisation.
                         Beside it the scalar instructions per vector instruction (they take issue time too) and the
                         share of wave-cycles spent waiting.
                 "l1"    the CU's vector memory path (address unit TA, L1 tags TCP, data return TD) serving hits: frac =
                         TCP_TOTAL_CACHE_ACCESSES / (CUs x cycles) / the MEASURED rate of the kernel's load shape from an
                         L1-resident table (profiles/r03_gather_rate.json: 1.4 accesses per clock and CU), TA / TD busy beside it.
                 "fill"  the filling of the L1's misses: 128-byte lines requested from the L2 per second against the
                         MEASURED rates of random line fills (same table), from the L2 for the share that hits there and
                         from beyond it for the rest.
               Without a matching profile: the line look-ups of the node and leaf fetches, counted live by the
               instrumented kernel (a lower bound of the L1's accesses).  Also reported: the HBM side (FETCH_SIZE x the
               factor calibrated on a one-line-per-lane gather, / launch time, / 8 TB/s) and the algorithmic bytes per ray
               (node visits x 64 B + leaf visits x 32 B of guard box + primitive tests x 48 B + 40 B, counted live at the
               timed resolution), most of which the caches serve.  Launch durations are HIP events on the launch stream.
  cpu_baseline the CPU oracle (a restatement of the reference's renderer, not the Rust binary)
               on a bounded sample of the same workload: same scene, samples and bounces at a
               reduced resolution, on this box's host cores (N=1 only)
  --total-passes P  strong scaling: P passes in all, split round-robin over the ranks ("scaling": "strong")

Launching: under a launcher (torchrun sets WORLD_SIZE / RANK / LOCAL_RANK) this process is one rank.  Without one,
`python bench.py --gpus N` with N > 1 starts the N ranks itself (one child process per GPU, rendezvous on 127.0.0.1)
before anything touches the GPU, relays rank 0's line and fails if any rank fails.  Either way the line is only
printed if the ranks that took part are the `--gpus` that were asked for ("ranks_seen", counted by an all-reduce).
"""
import argparse
import glob
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from srchash import kernel_source_hash  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
N_SIMD = 1024          # 256 CUs x 4
N_CU = 256
MAX_CLOCK_GHZ = 2.4


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


def config_key(args):
    """Names the timed configuration; a counter profile is only used for the configuration it was taken on."""
    return _config_key(args) + (" nodes=wide" if os.environ.get("RTCORE_NODES") == "wide" else "")


def _config_key(args):
    if args.workload == "closest_hit":
        return "closest_hit tris=%d rays=%d%s%s" % (args.tris, args.rays, " builder=device" if args.builder == "device" else "",
                                                     "" if args.inflate == 1.5 else " inflate=%g" % args.inflate)
    tris = args.tris if args.workload == "mesh" else 0
    return "%s tris=%d %dx%d spp=%d bounces=%d batch=%d%s" % (args.workload, tris, args.width, args.height, args.spp,
                                                             args.bounces, args.batch_paths,
                                                             " builder=device" if args.builder == "device" else "")


def matching_profile(key):
    """A counter profile under profiles/ taken from THIS kernel source (hash) on THIS configuration, or None."""
    want = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json")), reverse=True):
        try:
            p = json.load(open(path))
        except Exception:
            continue
        line = p.get("bench_line_under_rocprof") or {}
        e = p.get("extend") or {}
        if (p.get("source_hash") == want and (line.get("config") or {}).get("key") == key
                and "valu_busy_fraction" in e and "valu_lane_utilisation" in e and "hbm_bytes_per_launch_raw" in e):
            p["_file"] = os.path.relpath(path, ROOT)
            return p
    return None


def fetch_factor():
    """Bytes per FETCH_SIZE byte on a one-128-B-line-per-lane random gather (tools/ubench/gather_fetch.hip), the node
    fetch pattern of the kernel; 1.0 with a note when no calibration is committed."""
    path = os.path.join(ROOT, "profiles", "r02_fetch_calibration.json")
    if os.path.exists(path):
        c = json.load(open(path))
        if "gather128" in c:
            return (round(c["gather128"]["factor_vs_lines"], 3),
                    "calibrated on %s: known bytes of a one-128-B-line-per-lane random gather / FETCH_SIZE" % os.path.relpath(path, ROOT))
    return 1.0, "uncalibrated (raw FETCH_SIZE x 1 KiB)"


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, command, env=None, timeout=None):
    """Runs `command` (a list) as n rank processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set,
    relays rank 0's stdout (the other ranks' stdout is dropped, every rank's stderr passes through) and returns the
    exit status: 0 only if every rank ended with 0.  When a rank fails the others are ended (their own PIDs only)."""
    import subprocess

    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                 "LOCAL_WORLD_SIZE": str(n)})
    procs = []
    for r in range(n):
        e = dict(base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen(command, env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    t_end = None if timeout is None else time.time() + timeout
    status = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                sys.stderr.write("bench.py: rank %d ended with status %d; ending the other ranks\n" % (r, rc))
        if live and (status != 0 or (t_end is not None and time.time() > t_end)):
            if status == 0:
                status = 124
                sys.stderr.write("bench.py: ranks still running after %.0f s; ending them\n" % timeout)
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        if live:
            time.sleep(0.05)
    return status


def valu_ceiling(waves_per_simd):
    """The measured vector issue ceiling (profiles/r03_valu_rate.json, made by tools/ubench/valu_rate.hip on MI355X): cycles
    per vector instruction and SIMD of the node step's instruction mix at `waves_per_simd`; the guide's 2 cycles per
    wave64 instruction (SIMD-32) when no table is committed."""
    path = os.path.join(ROOT, "profiles", "r03_valu_rate.json")
    if os.path.exists(path):
        t = json.load(open(path))
        by_w = t["node_step"]["cycles_per_vector_instruction_and_simd_by_waves_per_simd"]
        w = str(waves_per_simd) if str(waves_per_simd) in by_w else sorted(by_w, key=lambda k: abs(int(k) - waves_per_simd))[0]
        mix = t["rows"].get("v_fma_f32 + s_and_b64 (2 : 1)", {}).get("cycles_per_instruction_and_simd_by_waves_per_simd", {})
        return {"cycles_per_vector_instruction": by_w[w], "cycles_per_vector_instruction_2_to_1_scalar": mix.get(w),
                "source": "profiles/r03_valu_rate.json: node step (rtd::qnode_children_sorted, %d vector instructions) at %s waves per SIMD, "
                          "in-kernel s_memtime cycles" % (t["node_step"]["loop_instructions"]["vector"], w)}
    return {"cycles_per_vector_instruction": 2.0, "cycles_per_vector_instruction_2_to_1_scalar": None,
            "source": "no measured table committed: 2 cycles per wave64 vector instruction (SIMD-32)"}


def gather_ceiling():
    """The measured rates of the CU's vector memory path for the kernel's load shape (profiles/r03_gather_rate.json, made by
    tools/ubench/quad_fetch.hip on MI355X): per-lane 16-byte loads that hit in the L1 (accesses per clock and CU, every lane
    active and on a line of its own), and the whole chip's rate of 128-byte line fills from the L2 and from beyond it.  One
    access per clock and CU and no fill rates when no table is committed."""
    path = os.path.join(ROOT, "profiles", "r03_gather_rate.json")
    if os.path.exists(path):
        t = json.load(open(path))
        return {"l1_accesses_per_clk_and_cu": t["l1_hit"]["accesses_per_clk_and_cu_all_lanes"],
                "fill_from_l2_G_per_s": t["line_fill_G_per_s"]["from_l2"],
                "fill_beyond_l2_G_per_s": t["line_fill_G_per_s"]["beyond_l2_2GB_table"],
                "source": "profiles/r03_gather_rate.json: four 16-byte loads per lane, every lane another node, 7 blocks per CU"}
    return {"l1_accesses_per_clk_and_cu": 1.0, "fill_from_l2_G_per_s": None, "fill_beyond_l2_G_per_s": None,
            "source": "no measured table committed: one line access per clock and CU"}


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
    ap.add_argument("--inflate", type=float, default=1.5,
                    help="closest_hit workload: origins are uniform in the scene's box scaled by this about its centre "
                         "(1.5 = the configuration's definition, many rays miss the room; 0.7 = every ray starts among the meshes)")
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--builder", default="exact", choices=["exact", "device"],
                    help="exact: the reference's tree, built on the host (default); device: the opt-in build on the GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--total-passes", type=int, default=0,
                    help="strong scaling: this many passes in all, split over the ranks (overrides --steps)")
    ap.add_argument("--reduce", default="auto", choices=["auto", "torch", "native"],
                    help="how the accumulators are summed at N > 1: torch.distributed (RCCL) or the library's own RCCL reduce")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    # one process per GPU; RT_BENCH_BACKEND=gloo lets several ranks share one card to rehearse the N > 1 control flow
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process becomes one and starts the ranks (nothing here has touched the GPU yet)
        import torch

        have = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
        if backend == "nccl" and have < args.gpus:
            raise SystemExit("bench.py: --gpus %d asked for, %d GPU(s) visible (RT_BENCH_BACKEND=gloo rehearses the "
                             "N > 1 control flow with the ranks sharing a card)" % (args.gpus, have))
        raise SystemExit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # the line's n_gpus must be what was asked for: a launcher that started another number of ranks is an error
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the core has no CPU path")
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
    desc.to_desc()  # (the flat description is made once more inside Scene; this call warms the arrays it points into)
    t0 = time.time()
    scene = pkg.Scene(desc, builder=pkg._abi.RT_BUILD_DEVICE if args.builder == "device" else pkg._abi.RT_BUILD_EXACT)
    t_build = time.time() - t0
    info = scene.info()
    stream = torch.cuda.current_stream().cuda_stream

    W, H, spp, bounces = args.width, args.height, args.spp, args.bounces
    strong = args.total_passes > 0 and args.workload != "closest_hit"
    # passes this rank renders in the timed region: rank, rank + world, ... (< total) when the total is fixed
    my_steps = len(range(rank, args.total_passes, world)) if strong else args.steps

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
        c, h = torch.tensor((mn + mx) / 2, device="cuda"), torch.tensor((mx - mn) / 2 * args.inflate, device="cuda")
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
            # one pass of the timed configuration through the counting kernel
            p = pkg.Scene.params(W, H, spp, bounces, pass_index=10_000, seed=1234, batch_paths=args.batch_paths)
            tmp = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
            scene.set_counting(True)
            scene.render_pass_device(tmp.data_ptr(), p, stream)
            torch.cuda.synchronize()
            scene.set_counting(False)

    # ---- node visits / primitive tests per ray and the steps' lane occupancy, from the instrumented kernel (untimed)
    scene.stats()
    count_step()
    cs = scene.stats()
    visits_per_ray = cs.node_visits / max(1, cs.rays)
    tests_per_ray = cs.prim_tests / max(1, cs.rays)
    # compact nodes (64 B) come with a 32-byte guard box per leaf visit; the 128-byte nodes hold the guards themselves
    guard_bytes_per_ray = 32.0 * cs.leaf_step_lanes / max(1, cs.rays) if info.node_bytes == 64 else 0.0
    bytes_per_ray = visits_per_ray * info.node_bytes + guard_bytes_per_ray + tests_per_ray * info.primitive_bytes + 32 + 8
    lookups_per_ray = cs.l1_lookups / max(1, cs.rays)
    steps_counted = cs.node_steps + cs.leaf_steps
    step_lane_occupancy = ((cs.node_step_lanes + cs.leaf_step_lanes) / (64.0 * steps_counted)) if steps_counted else None

    # ---- the sum of the accumulators: the one collective of a render
    use_native = world > 1 and args.workload != "closest_hit" and args.reduce == "native"

    def reduce_sum(buf):
        if use_native:
            pkg.parallel.native_reduce(buf, rank, world, stream)
        elif backend == "nccl":
            dist.reduce(buf, dst=0)
        else:
            # rehearsal on another backend (gloo has no reduce for device tensors): through the host
            host = buf.cpu()
            dist.reduce(host, dst=0)
            buf.copy_(host)

    if use_native:
        pkg.parallel.native_comm_init(rank, world, dist)

    # ---- warmup, then the timed region
    for i in range(args.warmup):
        step(i)
    if world > 1 and args.workload != "closest_hit" and args.warmup > 0:
        # the warmup also runs the collective once, on a scratch buffer of the accumulator's size, so that
        # the communicator's channels and buffers exist before the timed region
        scratch = torch.zeros_like(accum)
        reduce_sum(scratch)
        del scratch
    barrier()
    scene.stats()
    scene.set_stage_timing(True)
    t_start = time.perf_counter()
    for i in range(my_steps):
        step(args.warmup + i)
    reduce_ms = None
    if world > 1 and args.workload != "closest_hit":
        torch.cuda.synchronize()
        t_red = time.perf_counter()
        reduce_sum(accum)
        torch.cuda.synchronize()
        reduce_ms = (time.perf_counter() - t_red) * 1e3
    barrier()
    elapsed = time.perf_counter() - t_start
    st = scene.stats()
    scene.set_stage_timing(False)

    t_all = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    r_all = torch.tensor([float(st.rays)], dtype=torch.float64, device="cuda")
    seen = torch.ones(1, dtype=torch.float64, device="cuda")  # every rank that takes part adds one
    red = torch.tensor([reduce_ms or 0.0], dtype=torch.float64, device="cuda")
    if world > 1:
        if backend != "nccl":  # (gloo reduces host tensors)
            t_all, r_all, seen, red = t_all.cpu(), r_all.cpu(), seen.cpu(), red.cpu()
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(r_all, op=dist.ReduceOp.SUM)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
    elapsed_max = float(t_all.item())
    rays_total = float(r_all.item())
    ranks_seen = int(round(float(seen.item())))
    if ranks_seen != args.gpus:
        raise SystemExit("bench.py: %d rank(s) took part, --gpus %d was asked for" % (ranks_seen, args.gpus))

    if rank == 0:
        value = rays_total / elapsed_max / 1e6
        steps_reported = args.total_passes if strong else args.steps
        extend_s = st.extend_ms / 1e3
        launches = max(1, int(st.extend_launches))
        key = config_key(args)
        prof = matching_profile(key)
        gather = gather_ceiling()
        roof = {
            "kernel": "extend_kernel (closest hit)",
            "launches": launches,
            "mean_launch_ms": round(st.extend_ms / launches, 3),
            "extend_Mrays_per_s": round(st.rays / extend_s / 1e6, 1) if extend_s > 0 else None,
            "extend_ms_total": round(st.extend_ms, 3),
            "shade_ms_total": round(st.shade_ms, 3),
            "other_ms_total": round(st.other_ms, 3),
            "step_lane_occupancy_live": round(step_lane_occupancy, 4) if step_lane_occupancy is not None else None,
            "node_step_lanes_live": round(cs.node_step_lanes / max(1, cs.node_steps), 1),
            "leaf_step_lanes_live": round(cs.leaf_step_lanes / max(1, cs.leaf_steps), 1),
            "algorithmic_bytes_per_ray": round(bytes_per_ray, 1),
            "algorithmic_bytes_per_launch": int(st.rays * bytes_per_ray / launches),
            "algorithmic_GBps": round(st.rays * bytes_per_ray / extend_s / 1e9, 1) if extend_s > 0 else None,
            "node_visits_per_ray": round(visits_per_ray, 2),
            "prim_tests_per_ray": round(tests_per_ray, 2),
        }
        if extend_s > 0 and lookups_per_ray > 0:
            # A wave's load instruction costs the vector L1 one look-up per DIFFERENT line among its lanes; the peak is
            # the measured rate of this load shape from an L1-resident table.  Look-ups of the node and leaf fetches are
            # counted live by the instrumented kernel; ray loads, hit stores and stack overflow are not included.
            peak = N_CU * MAX_CLOCK_GHZ * gather["l1_accesses_per_clk_and_cu"]
            achieved = st.rays * lookups_per_ray / extend_s / 1e9
            roof["l1_lookup"] = {"achieved_G_per_s": round(achieved, 1), "peak_G_per_s": round(peak, 1),
                                 "frac": round(achieved / peak, 4), "lookups_per_ray": round(lookups_per_ray, 1),
                                 "note": "%.2f line look-ups per clock and CU (%s); counted for node and leaf fetches only"
                                         % (gather["l1_accesses_per_clk_and_cu"], gather["source"])}
        if prof is not None:
            e = prof["extend"]
            clock = e.get("clock_GHz_under_profile") or MAX_CLOCK_GHZ
            prof_launch_s = e["mean_launch_ms"] / 1e3
            cu_cycles = prof_launch_s * clock * 1e9 * N_CU  # CU-cycles of one launch
            simd_cycles = cu_cycles * 4
            # Vector ALU.  The ceiling is MEASURED (tools/ubench/valu_rate.hip, cycles from s_memtime inside the kernel):
            # cycles per vector instruction and SIMD of the node step's own code on register-resident nodes, all 64
            # lanes active, no memory access, at the waves per SIMD the kernel runs with.  peak = every SIMD issuing at
            # that rate with every lane useful; achieved = the kernel's vector instructions x active lanes per
            # instruction; frac = issue_frac x lane utilisation.
            ceil = valu_ceiling(int(e.get("waves_per_simd", 7)))
            valu_insts, salu_insts = e["SQ_INSTS_VALU_per_launch"], e["SQ_INSTS_SALU_per_launch"]
            cycles_per_valu = simd_cycles / valu_insts
            issue_frac = ceil["cycles_per_vector_instruction"] / cycles_per_valu
            lane_util = e["valu_lane_utilisation"]
            valu_peak = N_SIMD * 64 * clock / ceil["cycles_per_vector_instruction"]  # G lane-op/s
            valu_achieved = valu_insts * 64 * lane_util / prof_launch_s / 1e9
            factor, factor_note = fetch_factor()
            hbm_bytes = e["hbm_bytes_per_launch_raw"] * factor
            valu = {"achieved_G_lane_op_per_s": round(valu_achieved, 1), "peak_G_lane_op_per_s": round(valu_peak, 1),
                    "frac": round(issue_frac * lane_util, 4), "issue_frac": round(issue_frac, 4),
                    "valu_lane_utilisation": round(lane_util, 4),
                    "cycles_per_vector_instruction_and_simd": round(cycles_per_valu, 3),
                    "ceiling_cycles_per_vector_instruction_and_simd": ceil["cycles_per_vector_instruction"],
                    "ceiling_source": ceil["source"],
                    "scalar_per_vector_inst": round(salu_insts / valu_insts, 3),
                    # scalar instructions take issue time too (a stream of two v_fma_f32 to one s_and_b64 and its no-ops
                    # issues a vector instruction every so many cycles): context for scalar_per_vector_inst, not the ceiling
                    "cycles_per_vector_instruction_in_a_2_to_1_vector_scalar_stream": ceil.get("cycles_per_vector_instruction_2_to_1_scalar"),
                    "rocprof_valu_busy": round(e["valu_busy_fraction"], 4),
                    "wait_any_frac_of_wave_cycles": round(e["SQ_WAIT_ANY_per_launch"] / e["SQ_WAVE_CYCLES_per_launch"], 4)
                    if e.get("SQ_WAIT_ANY_per_launch") and e.get("SQ_WAVE_CYCLES_per_launch") else None}
            hbm = {"achieved_GBps": round(hbm_bytes / prof_launch_s / 1e9, 1), "peak_GBps": HBM_PEAK_GBS,
                   "frac": round(hbm_bytes / prof_launch_s / 1e9 / HBM_PEAK_GBS, 4),
                   "fetch_size_raw_bytes": int(e["hbm_bytes_per_launch_raw"]), "factor": factor, "factor_note": factor_note,
                   "l2_hit_rate": round(e.get("l2_hit_rate", 0.0), 4)}
            roof.update({"traffic": int(hbm_bytes), "valu": valu, "hbm": hbm,
                         "profile": {"file": prof["_file"], "source_hash": prof["source_hash"],
                                     "mean_launch_ms_under_profile": round(e["mean_launch_ms"], 3),
                                     "note": "counters from separate rocprofv3 --pmc runs of this configuration (one run per counter group), "
                                             "not from this run"}})
            acc = e.get("TCP_TOTAL_CACHE_ACCESSES_sum_per_launch")
            l1 = fill = None
            if acc:
                # per-lane 16-byte gathers load the CU's vector memory path (address unit TA -> L1 tags TCP -> data
                # return TD); the peak is what the same load shape reaches from an L1-resident table with every lane active
                l1_peak = gather["l1_accesses_per_clk_and_cu"]
                l1 = {"achieved_G_line_accesses_per_s": round(acc / prof_launch_s / 1e9, 1),
                      "peak_G_line_accesses_per_s": round(N_CU * clock * l1_peak, 1),
                      "frac": round(acc / cu_cycles / l1_peak, 4), "accesses_per_launch": int(acc),
                      "accesses_per_clk_and_cu": round(acc / cu_cycles, 4), "peak_accesses_per_clk_and_cu": l1_peak,
                      "ta_busy": round(e.get("TA_TA_BUSY_sum_per_launch", 0.0) / cu_cycles, 4),
                      "td_busy": round(e.get("TD_TD_BUSY_sum_per_launch", 0.0) / cu_cycles, 4),
                      "td_waiting_for_l1": round(e.get("TD_TC_STALL_sum_per_launch", 0.0) / cu_cycles, 4),
                      "ta_stalled_by_l1": round(e.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum_per_launch", 0.0) / cu_cycles, 4),
                      "peak_source": gather["source"],
                      "note": "TCP_TOTAL_CACHE_ACCESSES / (CUs x cycles) against the measured rate of L1 hits; TA / TD busy = the "
                              "address and data-return units' busy cycles / cycles (TD busy includes its waiting for the L1)"}
                roof["l1"] = l1
                req = e.get("TCP_TCC_READ_REQ_sum_per_launch")
                if req and gather["fill_from_l2_G_per_s"]:
                    # the L1's misses: 128-byte lines filled from the L2 (its hit rate) or from beyond it; the peak is the
                    # harmonic mix of the two measured fill rates at the kernel's L2 hit rate
                    h = e.get("l2_hit_rate", 0.0)
                    peak = 1.0 / (h / gather["fill_from_l2_G_per_s"] + (1.0 - h) / gather["fill_beyond_l2_G_per_s"])
                    achieved = req / prof_launch_s / 1e9
                    fill = {"achieved_G_lines_per_s": round(achieved, 1), "peak_G_lines_per_s": round(peak, 1),
                            "frac": round(achieved / peak, 4), "requests_to_l2_per_launch": int(req), "l2_hit_rate": round(h, 4),
                            "fill_from_l2_G_per_s": gather["fill_from_l2_G_per_s"],
                            "fill_beyond_l2_G_per_s": gather["fill_beyond_l2_G_per_s"], "peak_source": gather["source"],
                            "note": "L1 misses (TCP_TCC_READ_REQ, 128-byte lines) per second against the measured fill rates of "
                                    "random lines: from the L2 for the share that hits there, from beyond it (2 GB table) for the rest"}
                    roof["fill"] = fill
            # the busiest unit names the bound: the SIMDs' vector issue (against the measured ceiling), the L1's hit path or
            # the filling of its misses (against the measured gather rates)
            cands = [("valu", issue_frac)]
            if l1 is not None:
                cands.append(("l1", l1["frac"]))
            if fill is not None:
                cands.append(("fill", fill["frac"]))
            bound = max(cands, key=lambda c: c[1])[0]
            if bound == "fill":
                roof.update({"bound": "fill", "unit": "G lines/s", "achieved": fill["achieved_G_lines_per_s"],
                             "peak": fill["peak_G_lines_per_s"], "frac": fill["frac"]})
            elif bound == "l1":
                roof.update({"bound": "l1", "unit": "G line accesses/s", "achieved": l1["achieved_G_line_accesses_per_s"],
                             "peak": l1["peak_G_line_accesses_per_s"], "frac": l1["frac"]})
            else:
                roof.update({"bound": "valu", "unit": "G lane-op/s", "achieved": valu["achieved_G_lane_op_per_s"],
                             "peak": valu["peak_G_lane_op_per_s"], "frac": valu["frac"]})
        else:
            # no counter profile of this code and configuration: the look-ups counted live by the instrumented kernel
            # (node and primitive fetches only: a lower bound of the L1's accesses)
            lk = roof.get("l1_lookup")
            roof.update({"bound": "l1", "unit": "G line accesses/s",
                         "achieved": lk["achieved_G_per_s"] if lk else None, "peak": lk["peak_G_per_s"] if lk else None,
                         "frac": lk["frac"] if lk else None, "traffic": None,
                         "profile": {"file": None,
                                     "note": "no profile under profiles/ matches source hash %s and configuration '%s'; achieved is "
                                             "the live count of the node and primitive fetches' line look-ups, a lower bound of the "
                                             "L1's accesses" % (kernel_source_hash(), key)}})
        out = {
            "metric": "Mrays/s (extend+shade)",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": steps_reported,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / max(1, steps_reported) * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl_name,
                "kind": args.workload,
                "key": key,
                "width": W, "height": H, "spp_per_pass": spp, "max_bounce": bounces,
                "primitives": int(info.n_primitives), "wide_nodes": int(info.n_wide_nodes),
                "scene_MB": round(info.device_bytes / 1e6, 1),
                "parallelism": ("%d passes in all, " % args.total_passes if strong else "") +
                               "passes round-robin over %d GPU(s), scene replicated, %s" % (
                                   world, "no collective (one rank)" if world == 1 else
                                   "one RCCL sum (the library's ncclReduce)" if use_native else
                                   "one RCCL sum (torch.distributed.reduce)" if backend == "nccl" else
                                   "one sum through the host (%s back end: a rehearsal, not RCCL)" % backend),
                "reduce_ms": round(float(red.item()), 3) if world > 1 and args.workload != "closest_hit" else None,
                "s_per_4k_frame": round(elapsed_max / max(1, my_steps), 4) if (W, H) == (3840, 2160) else None,
                "rays_per_step_per_gpu": int(st.rays / max(1, my_steps)),
                "scene_gen_s": round(t_gen, 2), "scene_build_s": round(t_build, 2),
                "builder": args.builder, "node_bytes": int(info.node_bytes), "scene_build_ms_library": round(info.build_ms, 1), "wide_depth": int(info.wide_depth),
            },
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(pkg, desc, args, W, H, spp, bounces, info)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(pkg, desc, args, W, H, spp, bounces, info):
    """The CPU oracle on a bounded sample of the workload (checker code, timed here as the CPU column)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    # the GPU box grants a one-GPU job 16 host CPUs (it exports OMP_NUM_THREADS = 16); the machine itself has more
    host_cpus = len(os.sched_getaffinity(0))
    threads = min(host_cpus, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    ob.lib().oracle_set_counting(0)  # per-node counters off: this leg is timed
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
        return {"value": round(n * reps / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "host_cpus_visible": host_cpus,
                "kind": "port", "sample": "%d x %d random rays, same scene" % (reps, n)}
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
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "host_cpus_visible": host_cpus,
            "kind": "port",
            "sample": "one pass of the same scene, camera, %d spp, %d bounces at %dx%d (%.1f s)" % (spp, bounces, w1, h1, dt)}


if __name__ == "__main__":
    main()
