"""Sharding of a render over the GPUs of a node: one process per GPU, scene replicated.

Passes are independent and additive (src/render.rs:214, the final division by the pass count is
src/main.rs:288), and every random number is a function of the global pass index, so rank r of N
renders passes r, r + N, r + 2N, ... into its own accumulator and ONE sum-reduce (RCCL over xGMI when
the tensors are on GPUs, gloo on the CPU) gives the same frame as a single process that renders all
passes, up to the order of the float additions across ranks.

Pixels are independent within a pass as well (src/render.rs:163-176), so a single pass can be split by
pixel ranges instead: every rank renders its rows into a zeroed frame and the same sum-reduce puts the
frame together, bit for bit what one process renders (adding zeros is exact).
"""


def passes_for_rank(rank, world, total_passes):
    """Global pass indices rank `rank` of `world` renders."""
    return list(range(rank, total_passes, world))


def render_sharded(render_pass, accum, total_passes, rank=0, world=1, reduce_to=0, dist=None):
    """render_pass(accum, pass_index) adds one pass into `accum` (a torch tensor, any device).

    Returns the number of passes this rank rendered.  After the call rank `reduce_to` holds the sum
    over all ranks (other ranks' buffers are unspecified), the caller divides by total_passes.
    """
    mine = passes_for_rank(rank, world, total_passes)
    for p in mine:
        render_pass(accum, p)
    if world > 1:
        if dist is None:
            import torch.distributed as dist
        dist.reduce(accum, dst=reduce_to)
    return len(mine)


def pixels_for_rank(rank, world, n_pixels):
    """(first_pixel, pixel_count) of rank `rank`: contiguous, near-equal ranges that cover [0, n_pixels)."""
    first = n_pixels * rank // world
    return first, n_pixels * (rank + 1) // world - first


def render_split_frame(render_range, accum, n_pixels, rank=0, world=1, reduce_to=0, dist=None):
    """render_range(accum, first_pixel, pixel_count) adds this rank's pixels of ONE pass into `accum`, which
    must be zero outside that range.  After the call rank `reduce_to` holds the whole frame."""
    first, count = pixels_for_rank(rank, world, n_pixels)
    if count:
        render_range(accum, first, count)
    if world > 1:
        if dist is None:
            import torch.distributed as dist
        dist.reduce(accum, dst=reduce_to)
    return first, count


# ---- the library's own RCCL reduce, for one process per GPU ------------------------------------------------
_comm = None


def native_comm_init(rank, world, dist=None):
    """Builds the library's communicator (rt_comm_init_rank): rank 0 draws the id, the caller's process group (any
    backend) hands its 128 bytes round.  The current device must be this rank's GPU."""
    import ctypes as C

    import torch

    from . import _abi

    global _comm
    if _comm is not None:
        return _comm
    if dist is None:
        import torch.distributed as dist
    lib = _abi.load()
    ident = (C.c_ubyte * 128)()
    if rank == 0:
        _abi.check(lib.rt_comm_unique_id(ident))
    t = torch.tensor(list(ident), dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    ident = (C.c_ubyte * 128)(*t.cpu().tolist())
    h = C.c_void_p()
    _abi.check(lib.rt_comm_init_rank(ident, rank, world, C.byref(h)))
    _comm = h
    return h


def native_reduce(accum, rank, world, stream=0, root=0):
    """Sum of the ranks' device tensors `accum` into rank `root`'s, in place, enqueued on `stream`."""
    from . import _abi

    assert _comm is not None, "native_comm_init first"
    _abi.check(_abi.load().rt_comm_reduce(_comm, accum.data_ptr(), accum.numel(), root, stream))
