import sys
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene('/tmp/scn', width=960, height=540, spp=16)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
for mode in (False, True, False, True):
    ren.set_profiling(mode)
    t = []
    for k in range(4):
        img, st = ren.render(cam)
        t.append(round(st.kernel_ms, 2))
    print("profiling", mode, t)
