"""Sharding of a render over the GPUs of a node: one process per GPU, scene replicated.

Passes are independent and additive (src/render.rs:214, the final division by the pass count is
src/main.rs:288), and every random number is a function of the global pass index, so rank r of N
renders passes r, r + N, r + 2N, ... into its own accumulator and ONE sum-reduce (RCCL over xGMI when
the tensors are on GPUs, gloo on the CPU) gives the same frame as a single process that renders all
passes, up to the order of the float additions across ranks.
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
