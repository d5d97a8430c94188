"""Sample-index sharding of one render over the GPUs of a node.

Every (pixel, sample) is independent in the reference (src/renderer.rs:114-127);
the only combination step is the per-pixel sum (renderer.rs:123,128).  Rank r of
W therefore traces the sample indices [r*S/W, (r+1)*S/W) of EVERY pixel with the
same seed -- the stratified permutations are keyed on (seed, pixel, dimension)
only, so the shards together still visit each stratum exactly once -- and the
per-rank radiance sums are added with one reduce (RCCL over xGMI on GPUs, gloo
in the CPU tests).  One process per GPU; torch.distributed is plumbing only.
"""
import math


def shard_range(total_samples, rank, world_size):
    """Half-open sample-index range of `rank`; the ranges tile [0, total_samples)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    lo = (total_samples * rank) // world_size
    hi = (total_samples * (rank + 1)) // world_size
    return lo, hi


def weak_scaling_strata(x_strata, y_strata, world_size):
    """Strata grid for `world_size` times the samples of an x*y stratified config:
    factors of the multiplier are folded into x and y alternately (8 GPUs on
    128x128 -> 512x256), so x*y == world_size * x_strata * y_strata."""
    m = world_size
    x, y = x_strata, y_strata
    turn = 0
    f = 2
    while m > 1:
        while m % f:
            f += 1
        if turn == 0:
            x *= f
        else:
            y *= f
        turn ^= 1
        m //= f
    assert x * y == world_size * x_strata * y_strata
    return x, y


def reduce_sum(tensor, dst=0):
    """Sum `tensor` over all ranks into rank `dst` (no-op outside a process group)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    return tensor


def message_bytes(width, height):
    """Bytes of one rank's accumulator (float32 RGB)."""
    return width * height * 3 * 4
