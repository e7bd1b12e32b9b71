"""Tile sharding and the framebuffer gather without a GPU: the layout the library computes (who owns which tile, where
a tile lands in the root's receive buffer) and, with two processes over gloo, the exchange itself.

In the two-process test each rank takes the same full CPU image (rendered by the test oracle, standing in for the
device), packs the tiles it owns the way the device's resolve kernel does, rank 1 sends its buffer to rank 0 (what the
library does with one grouped ncclSend / ncclRecv) and rank 0 unpacks by the library's slot table.  Rank 0 must end up
with the single-rank image byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracer_amd import sharding  # noqa: E402


def skew(world):
    import math
    s = max(1, int(0.381966 * world + 0.5))
    while math.gcd(s, world) != 1:
        s += 1
    return s


@pytest.mark.parametrize("res,world", [((64, 64), 1), ((1920, 1080), 2), ((3840, 2160), 8), ((97, 33), 3), ((5, 7), 4), ((3840, 2160), 3)])
def test_owner_map_partitions_the_frame(res, world):
    own = sharding.owner_map(res[0], res[1], world)
    assert own.shape == ((res[1] + 31) // 32, (res[0] + 31) // 32)
    assert own.min() >= 0 and own.max() < world
    masks = np.stack([sharding.pixel_mask(res[0], res[1], r, world) for r in range(world)])
    assert (masks.sum(axis=0) == 1).all()                       # every pixel has exactly one owner
    # the rule: ranks follow each other along a row of tiles, shifted by a skew coprime to their number from row to row
    want = np.array([[(tx + skew(world) * ty) % world for tx in range(own.shape[1])] for ty in range(own.shape[0])])
    assert np.array_equal(own, want)
    if world > 1 and min(own.shape) >= world:                   # shares are even to within one row of tiles, and no rank owns a
        counts = np.bincount(own.ravel(), minlength=world)      # whole column or row of tiles
        assert counts.max() - counts.min() <= max(own.shape) // world + 1
        for r in range(world):
            assert not (own == r).all(axis=0).any() and not (own == r).all(axis=1).any()


@pytest.mark.parametrize("res", [(64, 64), (1920, 1080), (97, 33), (5, 7)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_gather_layout_is_a_bijection(res, world):
    """Every tile has exactly one slot in the root's buffer; rank r's slots follow those of ranks below it and keep the
    row-major order of the rank's own list."""
    counts, slot = sharding.layout(res[0], res[1], world)
    own = sharding.owner_map(res[0], res[1], world)
    assert counts.sum() == own.size and np.array_equal(counts, np.bincount(own.ravel(), minlength=world))
    assert np.array_equal(np.sort(slot.ravel()), np.arange(own.size))
    first = np.concatenate([[0], np.cumsum(counts)])
    for r in range(world):
        mine = slot.ravel()[own.ravel() == r]                   # row-major walk over the rank's tiles
        assert np.array_equal(mine, first[r] + np.arange(counts[r]))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_pack_then_unpack_is_the_frame(world):
    rng = np.random.default_rng(world)
    frame = rng.random((70, 100, 3), dtype=np.float32)          # 4 x 3 tiles, ragged on both axes
    gathered = np.concatenate([sharding.pack_tiles(frame, r, world) for r in range(world)])
    assert gathered.shape[0] == 12
    assert sharding.unpack_tiles(gathered, 100, 70, world).tobytes() == frame.tobytes()


def _worker(rank, world, port, full_path, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(full_path)
    h, w = full.shape[:2]
    counts, _ = sharding.layout(w, h, world)
    mine = torch.from_numpy(sharding.pack_tiles(full, rank, world))
    assert mine.shape[0] == counts[rank]
    if rank == 0:
        parts = [mine]
        for r in range(1, world):
            buf = torch.empty((int(counts[r]), 32, 32, 3), dtype=torch.float32)
            dist.recv(buf, src=r)
            parts.append(buf)
        np.save(out_path, sharding.unpack_tiles(torch.cat(parts).numpy(), w, h, world))
    else:
        dist.send(mine, dst=0)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_is_byte_identical(tmp_path):
    import raytracer_amd as R
    from oracle_lib import Oracle
    sc = R.Scene(os.path.join(ROOT, "scenes", "simple.json"))
    cam = sc.camera(0, resx=80, resy=72)                        # 3 x 3 tiles, ragged on both axes
    full, _ = Oracle(sc).render(cam, threads=2)
    assert full.std() > 0
    np.save(tmp_path / "full.npy", full)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path / "full.npy"), str(tmp_path / "out.npy")), nprocs=2, join=True)
    out = np.load(tmp_path / "out.npy")
    assert out.tobytes() == full.tobytes()


class _NoRccl:
    """Stands in for the GPU library on rank 0: the id cannot be made."""

    def rt_comm_unique_id(self, buf, n):
        return 11

    def rt_last_error(self):
        return b"rt_comm_unique_id: librccl.so could not be loaded"


def _id_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        sharding.gpu_lib = lambda: _NoRccl()
    try:
        sharding.exchange_unique_id(rank, world)
        said = "no error"
    except RuntimeError as e:
        said = str(e)
    # what bench.py does next: every rank enters the same collective
    ok = torch.tensor([0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(said)
    dist.destroy_process_group()


def test_failed_id_on_rank_zero_reaches_every_rank(tmp_path):
    """Rank 0 cannot make the communicator's id: the failure travels in the broadcast, every rank raises it and all of
    them meet again in the next collective (two gloo processes; it used to leave rank 1 in the broadcast for ever)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.spawn(_id_worker, args=(2, port, str(tmp_path)), nprocs=2, join=False)
    deadline = 120
    import time
    t0 = time.time()
    while not ctx.join(timeout=1):
        assert time.time() - t0 < deadline, "a rank hangs in the id exchange"
    for r in range(2):
        assert "librccl.so could not be loaded" in open(tmp_path / f"rank{r}.txt").read()


def test_queue_shards_get_an_even_share_and_the_planner_knows_the_fullest():
    """Inside a device the camera samples of a batch are dealt to 64 queue shards (workgroup q of sample s -> shard
    (q + 37 s) mod 64): every quarter tile visits every shard within 64 samples, a shard's share of a batch stays within one
    workgroup per sample of the mean, and rt_queue_fill_bound is exactly the fullest shard's count."""
    import ctypes as C
    import raytracer_amd as R
    g = R.gpu_lib()
    g.rt_queue_shard_of.restype = C.c_int32
    g.rt_queue_shard_of.argtypes = [C.c_uint32, C.c_uint32]
    g.rt_queue_fill_bound.restype = C.c_uint64
    g.rt_queue_fill_bound.argtypes = [C.c_uint32, C.c_uint32]
    assert sorted(g.rt_queue_shard_of(5, s) for s in range(64)) == list(range(64))
    rng = np.random.default_rng(7)
    cases = [(4, 16), (4, 64), (1632, 64), (2944, 256), (2948, 256), (100, 1), (63, 7), (65, 100), (2500, 16)]
    cases += [(int(a), int(b)) for a, b in zip(rng.integers(1, 400, 12), rng.integers(1, 300, 12))]
    for groups, samples in cases:
        q = np.arange(groups, dtype=np.int64)[None, :]
        s = np.arange(samples, dtype=np.int64)[:, None]
        fill = np.bincount(((q + 37 * s) % 64).ravel(), minlength=64)
        assert g.rt_queue_shard_of(groups - 1, samples - 1) == (groups - 1 + 37 * (samples - 1)) % 64
        assert g.rt_queue_fill_bound(groups, samples) == fill.max(), (groups, samples)
        assert fill.max() - fill.min() <= min(samples, 64), (groups, samples)
        if groups % 64 == 0 or samples % 64 == 0:
            assert fill.max() == fill.min()
