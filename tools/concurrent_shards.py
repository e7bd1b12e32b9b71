"""Does rendering two or three tile shards of one frame concurrently (separate scene objects, separate streams)
beat one render of the whole frame?  Evidence for/against pipelining batches."""
import sys, threading, time
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene('/tmp/scn', width=960, height=540, spp=16)
sc = R.Scene(p); cam = sc.camera(0)
for world in (1, 2, 3, 4):
    rens = [R.Renderer(sc) for _ in range(world)]
    outs = [np.zeros((cam.resy, cam.resx, 3), dtype=np.float32) for _ in range(world)]
    def work(r):
        rens[r].render(cam, out=outs[r], rank=r, world=world)
    for rep in range(3):
        ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        ms = (time.perf_counter() - t0) * 1e3
    print(f"{world} concurrent shard(s): {ms:.2f} ms wall for the frame")
    for r in rens: r.close()
