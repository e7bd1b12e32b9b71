"""Tile sharding and the framebuffer gather with two processes over gloo (no GPU).

Each rank takes the same full CPU image (rendered by the test oracle, standing in for the device), keeps only the
pixels of the tiles it owns according to rt_shard_owner_map -- what rt_render_shard writes -- and the ranks run
the product's gather.  Rank 0 must end up with the single-rank image byte for byte."""
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


@pytest.mark.parametrize("res,world", [((64, 64), 1), ((1920, 1080), 2), ((3840, 2160), 8), ((97, 33), 3), ((5, 7), 4)])
def test_owner_map_partitions_the_frame(res, world):
    own = sharding.owner_map(res[0], res[1], world)
    assert own.shape == ((res[1] + 31) // 32, (res[0] + 31) // 32)
    assert own.min() >= 0 and own.max() < world
    masks = np.stack([sharding.pixel_mask(res[0], res[1], r, world) for r in range(world)])
    assert (masks.sum(axis=0) == 1).all()                       # every pixel has exactly one owner
    counts = np.bincount(own.ravel(), minlength=world)
    assert counts.max() - counts.min() <= 1                     # tiles are dealt round-robin
    # round-robin in row-major tile order (what the device kernel inverts: tile = owned_index * world + rank)
    assert np.array_equal(own.ravel(), np.arange(own.size) % world)


def _worker(rank, world, port, full_path, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(full_path)
    h, w = full.shape[:2]
    mine = np.where(sharding.pixel_mask(w, h, rank, world)[:, :, None], full, np.float32(0))
    fb = torch.from_numpy(np.ascontiguousarray(mine))
    sharding.gather_frame(fb, world)
    if rank == 0:
        np.save(out_path, fb.numpy())
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
