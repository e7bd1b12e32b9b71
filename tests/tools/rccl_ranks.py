"""Child process of tests/test_gpu_sharding.py: `world` ranks as THREADS of this process, all on GPU 0, each with its own
communicator made by rt_comm_init_rank from one id -- the set-up of one process per GPU, with the threads standing for the
processes -- render frames through rt_render_gather's RCCL branch.  RT_RCCL_LIB names the stand-in library
(tests/tools/fake_rccl.cpp) that carries the data between the threads; a real RCCL communicator takes one rank per device.

usage: rccl_ranks.py <scene.json> <world> <failing rank or -1>      prints one JSON line"""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C

import numpy as np

import raytracer_amd as R
from raytracer_amd import sharding


def main():
    path, world, failing = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    assert os.environ.get("RT_RCCL_LIB"), "the stand-in library must be named"
    sc = R.Scene(path)
    cam = sc.camera(0)
    full, st = R.Renderer(sc).render(cam)
    uid = sharding.exchange_unique_id(0, 1)     # rank 0's id; "handed" to the other ranks by sharing the variable
    rens = [R.Renderer(sc) for _ in range(world)]
    comms = [None] * world
    frames = 3
    out = [None] * frames
    rays = [[0] * world for _ in range(frames)]
    errors = [[None] * world for _ in range(frames)]

    def work(rank):
        comms[rank] = R.Comm.init_rank(uid, rank, world)    # returns when every rank has joined
        for f in range(frames):
            bad = f == 1 and rank == failing
            c = sc.camera(0, num_samples=0) if bad else cam   # a camera the renderer refuses: this rank's render fails
            try:
                img, s = rens[rank].render_gather(comms[rank], c)
                rays[f][rank] = s.total_rays
                if rank == 0:
                    out[f] = img
            except R.RaytracerError as e:
                errors[f][rank] = str(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    stuck = [t.is_alive() for t in threads]
    res = {"stuck": any(stuck), "world": world,
           "frame_equal": [o is not None and o.tobytes() == full.tobytes() for o in out],
           "rays_equal": [sum(r) == st.total_rays for r in rays],
           "errors": errors}
    print(json.dumps(res), flush=True)
    if any(stuck):
        os._exit(3)   # threads parked in the exchange would keep the interpreter from ending
    for c in comms:
        c.close()


if __name__ == "__main__":
    main()
