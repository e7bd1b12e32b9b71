"""Study script (not a test; run by hand, a few minutes): both walks of the checker on ~3 M random rays per scene -- camera-like\nrays, rays leaving the points they hit, any-hit rays towards three targets -- for every fixture scene, the 70 K mesh and\nthe 1M-triangle scene; prints the number of rays on which the walks disagree (hit, shape or distance bits)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import raytracer_amd as R
from raytracer_amd import scenegen
from golden import scenes as GS
from oracle_lib import Oracle
import test_wide_tree as T
rng = np.random.default_rng(123)
names = list(GS.SCENES) + ['mesh70k', 'pt1m']
for name in names:
    if name == 'pt1m':
        sc = R.Scene(scenegen.write_path_traced_scene('/tmp/wide_walks_diff/p', width=64, height=64, spp=1))
    else:
        sc = T.scene_of(name, '/tmp/wide_walks_diff')
    orc = Oracle(sc)
    if orc.wide_check()[0] != 0:
        print(name, 'no wide tree or unsound', orc.wide_check()); continue
    tot = bad = 0
    for rep in range(3):
        prim, sec = T.rays_for(sc, orc, 300000, rng)
        for o, d in (prim, sec):
            a = orc.intersect_many(o, d); b = orc.intersect_many(o, d, wide=True)
            bad += int(((a[0] != b[0]) | (a[2].view(np.uint32) != b[2].view(np.uint32))).sum()); tot += len(o)
        o = sec[0]
        for target in ([0.3, 3.4, 0.8], [3, 5, 4], [-4, 3, 2]):
            d = np.array(target, np.float32) - o
            dist = np.linalg.norm(d, axis=1).astype(np.float32); d = (d / dist[:, None]).astype(np.float32)
            a = orc.intersect_many(o, d, stop_t=dist, any_hit=True, eps=1e-3); b = orc.intersect_many(o, d, stop_t=dist, any_hit=True, eps=1e-3, wide=True)
            bad += int(((a[0] != b[0]) | (a[2].view(np.uint32) != b[2].view(np.uint32))).sum()); tot += len(o)
    print(name, tot, 'rays, disagreements', bad, flush=True)
