import sys, time, threading
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_mesh_scene('/tmp/scn', n=188, width=1920, height=1080)
for k in range(3):
    t0 = time.perf_counter(); sc = R.Scene(p); print("load", round(1e3 * (time.perf_counter() - t0), 1), "ms", sc.timings())
def work():
    t0 = time.perf_counter(); sc = R.Scene(p); print("  concurrent load", round(1e3 * (time.perf_counter() - t0), 1), "ms", sc.timings())
ths = [threading.Thread(target=work) for _ in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
