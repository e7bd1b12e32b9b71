import sys, time, os
sys.path.insert(0,'.')
import raytracer_amd as R
from raytracer_amd import scenegen
d='/tmp/scn'
p=scenegen.write_mesh_scene(d)
sc=R.Scene(p); ren=R.Renderer(sc); cam=sc.camera(0)
for k in range(6):   # the scene gets its wide tree with the fourth frame
    img,st=ren.render(cam); print('C2 1080p', st.kernel_ms, st.total_ms, st.total_rays/st.kernel_ms/1e3,'Mrays/s')
p=scenegen.write_path_traced_scene(d,width=960,height=540,spp=16)
sc=R.Scene(p); ren=R.Renderer(sc); cam=sc.camera(0)
for k in range(2):
    img,st=ren.render(cam); print('C4 960x540x16', st.kernel_ms, st.total_ms, st.total_rays, st.total_rays/st.kernel_ms/1e3,'Mrays/s', st.as_dict())
