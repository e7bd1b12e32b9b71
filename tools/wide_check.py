"""Wide-tree walk against the two-box walk on the GPU: images, primary hit ids and distances of the fixture scenes
and of the two benchmark meshes (reduced).  The counted render walks the two-box records."""
import sys, os
os.environ.setdefault('RT_WIDE_NODES', '1')
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import raytracer_amd as R
from raytracer_amd import scenegen
from golden import scenes as GS

d = '/tmp/wide_check'
paths = {n: b(d) for n, (b, k) in GS.SCENES.items()}
paths['mesh70k_full'] = scenegen.write_mesh_scene(d + '/m')
paths['pt1m'] = scenegen.write_path_traced_scene(d + '/p', width=480, height=270, spp=4)
bad = 0
for name, p in paths.items():
    sc = R.Scene(p)
    for ci in range(sc.n_cameras):
        cam = sc.camera(ci)
        ren = R.Renderer(sc)
        s0, i0, t0, _ = ren.trace_primary(cam)
        a, sa = ren.render(cam)
        ren.set_counters(True)
        s1, i1, t1, _ = ren.trace_primary(cam)
        b, sb = ren.render(cam)
        ids = int((s0 != s1).sum() + (i0 != i1).sum())
        ts = int((t0.view(np.uint32) != t1.view(np.uint32)).sum())
        err = np.abs(a - b) / np.maximum(np.abs(b), 1 / 255)
        npx = int((err.max(axis=2) > 1e-5).sum())
        rays = (sa.primary_rays, sa.secondary_rays) == (sb.primary_rays, sb.secondary_rays)
        print(f"{name}[{ci}] {cam.resx}x{cam.resy}: id diffs {ids}, t diffs {ts}, pixels off {npx} (max rel {err.max():.3g}), ray counts equal {rays}", flush=True)
        if npx:
            ys, xs = np.nonzero(err.max(axis=2) > 1e-5)
            print("   at", list(zip(xs.tolist(), ys.tolist()))[:8])
        bad += (ids or ts or npx)
        ren.close()
print("FAIL" if bad else "OK")
