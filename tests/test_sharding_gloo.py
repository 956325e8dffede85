"""The N > 1 path on the CPU: world_size-2 gloo, passes sharded round-robin, one reduce.
The oracle stands in for the GPU renderer (same pass/seed contract), so this checks that
the sharded sum equals the single-process sum of the same passes."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, BOUNCES, PASSES, SEED = 48, 32, 2, 4, 5, 77


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge
    import oracle_binding as ob

    pkg = ge.load_package()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ora = ob.OracleScene(pkg.scenes.cornell())
    accum = torch.zeros(W * H * 3, dtype=torch.float32)

    def render_pass(buf, pass_index):
        p = pkg.Scene.params(W, H, SPP, BOUNCES, pass_index=pass_index, seed=SEED)
        ora.render_pass(buf.numpy(), p, threads=2)

    n = pkg.parallel.render_sharded(render_pass, accum, PASSES, rank, world, dist=dist)
    assert n == len(range(rank, PASSES, world))
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_render_equals_single_process(tmp_path, pkg, oracle):
    import torch.multiprocessing as mp

    out = str(tmp_path / "sharded.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sharded = np.load(out)
    ora = oracle.OracleScene(pkg.scenes.cornell())
    single = np.zeros(W * H * 3, np.float32)
    per_pass = []
    for i in range(PASSES):
        one = np.zeros(W * H * 3, np.float32)
        p = pkg.Scene.params(W, H, SPP, BOUNCES, pass_index=i, seed=SEED)
        ora.render_pass(one, p, threads=2)
        ora.render_pass(single, p, threads=2)
        per_pass.append(one)
    # identical up to the order of float additions across ranks
    assert np.allclose(sharded, single, rtol=1e-6, atol=1e-7)
    # and bit-identical to the sum taken in the sharded order: (p0+p2+p4) + (p1+p3)
    r0 = per_pass[0] + per_pass[2] + per_pass[4]
    r1 = per_pass[1] + per_pass[3]
    assert np.array_equal(sharded, r0 + r1)


def test_pass_partition(pkg):
    for world in (1, 2, 3, 8):
        seen = sorted(sum((pkg.parallel.passes_for_rank(r, world, 21) for r in range(world)), []))
        assert seen == list(range(21))


def _frame_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge
    import oracle_binding as ob

    pkg = ge.load_package()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ora = ob.OracleScene(pkg.scenes.cornell())
    accum = torch.zeros(W * H * 3, dtype=torch.float32)

    def render_range(buf, first, count):
        p = pkg.Scene.params(W, H, SPP, BOUNCES, pass_index=3, seed=SEED, first_pixel=first, pixel_count=count)
        ora.render_pass(buf.numpy(), p, threads=2)

    first, count = pkg.parallel.render_split_frame(render_range, accum, W * H, rank, world, dist=dist)
    assert (first, count) == pkg.parallel.pixels_for_rank(rank, world, W * H)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_one_pass_split_by_pixel_ranges_equals_single_process(tmp_path, pkg, oracle):
    """The single-pass alternative: the frame is cut into pixel ranges, one per rank; the reduce of the zero-padded
    pieces is bit for bit the frame one process renders."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "split.npy")
    mp.spawn(_frame_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    whole = np.zeros(W * H * 3, np.float32)
    oracle.OracleScene(pkg.scenes.cornell()).render_pass(whole, pkg.Scene.params(W, H, SPP, BOUNCES, pass_index=3, seed=SEED),
                                                         threads=2)
    assert np.array_equal(np.load(out).view(np.uint32), whole.view(np.uint32))


def test_pixel_partition(pkg):
    for world in (1, 2, 3, 8):
        for n in (1, 7, 48 * 32, 3840 * 2160):
            spans = [pkg.parallel.pixels_for_rank(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
