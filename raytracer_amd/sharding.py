"""Multi-GPU plumbing: one process per GPU, a frame is split by 32x32 tiles, rank 0 receives the whole frame.

Every rank renders only its own tiles (rt_render_shard) into a full-size, zero-initialised framebuffer, so the
gather is a single sum-reduce of disjoint tiles to rank 0 (RCCL on GPUs, gloo in the CPU tests).  Random numbers
are keyed by absolute pixel, so the gathered frame is byte-identical to a single-rank render: x + 0 == x in IEEE
arithmetic for every finite x (and -0.0 never occurs in a framebuffer of non-negative radiance sums).
"""
import numpy as np

from . import gpu_lib

TILE = 32


def owner_map(resx, resy, world):
    """(tiles_y, tiles_x) int32 array of owning ranks (tile row 0 = bottom image rows)."""
    tx, ty = (resx + TILE - 1) // TILE, (resy + TILE - 1) // TILE
    out = np.empty((ty, tx), dtype=np.int32)
    if gpu_lib().rt_shard_owner_map(resx, resy, world, out.ctypes.data) != 0:
        raise ValueError(gpu_lib().rt_last_error().decode())
    return out


def pixel_mask(resx, resy, rank, world):
    """(resy, resx) bool array: pixels rendered by `rank`."""
    own = owner_map(resx, resy, world) == rank
    return np.repeat(np.repeat(own, TILE, axis=0), TILE, axis=1)[:resy, :resx]


def gather_frame(fb, world):
    """Collects the ranks' disjoint tiles on rank 0, in place; `fb` is a torch tensor (any device)."""
    if world > 1:
        import torch.distributed as dist
        if fb.is_cuda and dist.get_backend() == "gloo":   # rehearsals without RCCL: stage through host memory
            host = fb.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if dist.get_rank() == 0:
                fb.copy_(host)
        else:
            dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM)
    return fb
