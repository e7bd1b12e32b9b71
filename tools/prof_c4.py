import os, sys
os.environ.setdefault("RT_WIDE_NODES", "1")
sys.path.insert(0,'.')
import raytracer_amd as R
from raytracer_amd import scenegen
p=scenegen.write_path_traced_scene('/tmp/scn',width=960,height=540,spp=16)
sc=R.Scene(p); ren=R.Renderer(sc); cam=sc.camera(0)
for k in range(3):
    img,st=ren.render(cam); print(st.kernel_ms, st.total_rays/st.kernel_ms/1e3)
