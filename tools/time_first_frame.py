import sys, time
sys.path.insert(0, ".")
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene("/tmp/scn_kt", width=3840, height=2160, spp=64)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
for k in range(3):
    t=time.time(); img, st = ren.render(cam); print("frame %d wall %.1f ms kernel %.1f" % (k, (time.time()-t)*1e3, st.kernel_ms), flush=True)
