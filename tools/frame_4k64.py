"""Two 4K / 64 spp frames of the 1M-triangle scene (for counter passes over a full-size frame)."""
import os, sys
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene('/tmp/scn_kt', width=3840, height=2160, spp=64)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
for k in range(2):
    img, st = ren.render(cam)
print(st.kernel_ms)
