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
