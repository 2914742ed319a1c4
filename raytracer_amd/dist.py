"""Multi-GPU partition rules of the render path (one process per GPU, torch.distributed over RCCL).

The path shards without any data-path collective inside the frame: every (pixel, sample) is independent.  Two orthogonal
partitions, both ending in ONE sum-reduce of the float4 accumulation buffer (sum r, sum g, sum b, divisor):
  * tile: 8x8-pixel tiles dealt round-robin, tile t belongs to rank t % world (exactly what libmirt does for
    mirt_options.tile_rank / tile_world); the other ranks contribute zeros, so the sum is bit-identical to one GPU;
  * spp : rank r renders samples [begin_r, begin_r + count_r) of every pixel (counter-based RNG only); float sums are
    re-associated, so the result matches one GPU to rounding.
"""
from __future__ import annotations

import numpy as np

TILE = 8


def tile_owner(width, height, world):
    """(h, w) int array: the rank that owns each pixel under the tile partition."""
    tx = np.arange(width) // TILE
    ty = np.arange(height) // TILE
    tiles_x = (width + TILE - 1) // TILE
    return ((ty[:, None] * tiles_x + tx[None, :]) % max(1, world)).astype(np.int32)


def spp_slices(spp, world):
    """[(begin, count)] per rank; the last rank takes the remainder.  Needs at least one sample per rank: libmirt reads
    sample_count == 0 as "all samples" (mirt.h), so an empty slice cannot be expressed."""
    if world < 1 or spp < world:
        raise ValueError(f"spp partition needs at least one sample per rank ({spp} spp, {world} ranks): use the tile partition")
    per = spp // world
    out = []
    for r in range(world):
        cnt = per if r < world - 1 else spp - per * (world - 1)
        out.append((r * per, cnt))
    return out


def partition_options(kind, rank, world, spp):
    """kwargs for capi.Scene.render_accum for this rank."""
    if world <= 1:
        return {}
    if kind == "tile":
        return dict(tile_rank=rank, tile_world=world)
    if kind != "spp":
        raise ValueError(f"unknown partition {kind!r}")
    begin, count = spp_slices(spp, world)[rank]
    return dict(rng_mode=1, sample_begin=begin, sample_count=count)


def reduce_accum(accum, dist, dst=0, via_host=False):
    """The frame's ONE collective: sum the per-rank accumulation buffers onto `dst` (RCCL reduce on GPUs; gloo in the CPU
    tests).  via_host: the ranks share GPUs (rehearsal on a box with fewer GPUs than ranks), so the buffers travel through
    host memory and gloo."""
    if via_host:
        host = accum.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            accum.copy_(host)
    else:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def render_partitioned(render_accum, accum, dist, rank, world, kind, spp, via_host=False, sync=None):
    """One frame over `world` ranks -- the N > 1 path of bench.py and of tests/test_dist_cpu.py: this rank renders its
    partition into `accum` (`render_accum(accum, **partition_options)` -> stats; pixels / samples it does not own contribute
    zeros), then the buffers are summed onto rank 0.  Returns the rank's stats; with N > 1 they carry `reduce_ms`, this
    rank's wall time in the collective (`sync`: waits for the device, e.g. torch.cuda.synchronize -- the RCCL reduce is
    asynchronous to the host; the wait also covers ranks that arrive late, so it is an upper bound of the transfer)."""
    stats = render_accum(accum, **partition_options(kind, rank, world, spp))
    if world > 1:
        import time

        t0 = time.perf_counter()
        reduce_accum(accum, dist, dst=0, via_host=via_host)
        if sync is not None:
            sync()
        stats = dict(stats)
        stats["reduce_ms"] = 1e3 * (time.perf_counter() - t0)
    return stats
