"""Per-kernel device time of the reduced path-traced workload (960x540, 16 spp, 1M triangles)."""
import sys
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene('/tmp/scn', width=960, height=540, spp=16)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
ren.set_profiling(True)
n = 3
for k in range(n):
    img, st = ren.render(cam)
kt = ren.kernel_times()
print("frame ms", st.kernel_ms, "Mrays/s", st.total_rays / st.kernel_ms / 1e3)
for k, (l, ms) in kt.items():
    print(f"  {k:14s} {l // n:4d} launches {ms / n:8.3f} ms/frame")
