"""Creates, renders once and destroys the 70K-triangle scene 60 times with the tree built on the device (what the
animation's render thread does per file); run under `rocprofv3 --hip-trace --stats` to see which runtime calls cost."""
import os, sys, time
os.environ["RT_GPU_BVH"] = "1"
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
path = scenegen.write_mesh_scene('/tmp/scn_cl', width=1920, height=1080)
sc = R.Scene(path)
cam = sc.camera(0)
for warm in range(3):
    ren = R.Renderer(sc); ren.render(cam); ren.close()
t_create = t_render = t_close = 0.0
n = 60
for k in range(n):
    t0 = time.perf_counter(); ren = R.Renderer(sc); t1 = time.perf_counter(); ren.render(cam); t2 = time.perf_counter(); ren.close(); t3 = time.perf_counter()
    t_create += t1 - t0; t_render += t2 - t1; t_close += t3 - t2
print("create %.2f ms, render (to host) %.2f ms, destroy %.2f ms per scene" % (t_create / n * 1e3, t_render / n * 1e3, t_close / n * 1e3))
