import sys, time
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_mesh_scene('/tmp/scn', n=188, width=1920, height=1080)
sc = R.Scene(p); cam = sc.camera(0)
for k in range(5):
    t0 = time.perf_counter(); ren = R.Renderer(sc); t1 = time.perf_counter()
    ren.render(cam); t2 = time.perf_counter()
    ren.render(cam); t3 = time.perf_counter()
    ren.close(); t4 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.1f} ms, first render {1e3*(t2-t1):.1f} ms, second render {1e3*(t3-t2):.1f} ms, destroy {1e3*(t4-t3):.1f} ms")
