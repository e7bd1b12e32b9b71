"""Multi-GPU plumbing around the library's tile-sharded frame (include/gpu_render.h, rt_render_gather).

A frame is split by 32x32 tiles; tile (tx, ty) belongs to rank (tx + skew * ty) mod world, the skew coprime to world.  Every rank renders its
tiles into a packed, tile-major buffer (its tiles in row-major order, 32*32*3 floats each, pixels row-major inside a
tile), the buffers go to rank 0 in one grouped send / receive (RCCL inside the library), and rank 0 moves every tile
to its place.  Random numbers are keyed by absolute pixel, so a pixel gets the same value whichever rank renders it.

This module holds the host-side views of that layout (who owns what, where a tile lands) and the hand-over of the
RCCL id between processes; `pack_tiles` / `unpack_tiles` restate on the host what the device's resolve and unpack
kernels do, for tests that run the exchange without a GPU.
"""
import ctypes as C

import numpy as np

from . import gpu_lib

TILE = 32
COMM_ID_BYTES = 128


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


def layout(resx, resy, world):
    """(tiles per rank [world], slot of every frame tile in rank 0's receive buffer [tiles_y, tiles_x])."""
    g = gpu_lib()
    g.rt_shard_layout.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    tx, ty = (resx + TILE - 1) // TILE, (resy + TILE - 1) // TILE
    counts = np.empty(world, dtype=np.int32)
    slot = np.empty((ty, tx), dtype=np.int32)
    if g.rt_shard_layout(resx, resy, world, counts.ctypes.data, slot.ctypes.data) != 0:
        raise ValueError(g.rt_last_error().decode())
    return counts, slot


def pack_tiles(frame, rank, world):
    """The packed buffer of `rank`: its tiles of `frame` ((resy, resx, 3) float32, row 0 = bottom row) in row-major tile
    order; pixels outside the image are zero."""
    resy, resx = frame.shape[:2]
    own = owner_map(resx, resy, world)
    tiles = []
    for ty, tx in zip(*np.nonzero(own == rank)):       # np.nonzero walks row-major
        t = np.zeros((TILE, TILE, 3), dtype=np.float32)
        part = frame[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
        t[:part.shape[0], :part.shape[1]] = part
        tiles.append(t)
    return np.stack(tiles) if tiles else np.zeros((0, TILE, TILE, 3), dtype=np.float32)


def unpack_tiles(gathered, resx, resy, world):
    """The frame from rank 0's receive buffer ((tiles, 32, 32, 3): rank after rank)."""
    _, slot = layout(resx, resy, world)
    frame = np.zeros((resy, resx, 3), dtype=np.float32)
    for ty in range(slot.shape[0]):
        for tx in range(slot.shape[1]):
            part = frame[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
            part[...] = gathered[slot[ty, tx], :part.shape[0], :part.shape[1]]
    return frame


def exchange_unique_id(rank, world):
    """The RCCL id of a communicator over the ranks of the current torch.distributed group: made by rank 0 inside the
    library, handed to the other ranks through torch.distributed (whatever backend it runs on).  When rank 0 cannot make
    the id (no RCCL to load) it says so IN the broadcast, so that every rank takes part in it and every rank raises the
    same error afterwards: no rank is left waiting in a collective the others never enter."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    error = None
    if rank == 0:
        g = gpu_lib()
        if g.rt_comm_unique_id(buf, COMM_ID_BYTES) != 0:
            error = g.rt_last_error().decode() or "rt_comm_unique_id failed"
    box = [{"error": error} if error else bytes(buf)]
    if world > 1:
        # (torch is imported only here: a process that has already brought up /opt/rocm's HIP runtime through the library
        # must not load the second ROCm stack that torch ships; launchers of several ranks import torch first)
        import torch.distributed as dist
        dist.broadcast_object_list(box, src=0)
    if isinstance(box[0], dict):
        raise RuntimeError("rank 0 could not make the communicator's id: " + box[0]["error"])
    return box[0]
