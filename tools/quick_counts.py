import sys
sys.path.insert(0,'.')
import raytracer_amd as R
from raytracer_amd import scenegen
p=scenegen.write_path_traced_scene('/tmp/scn',width=960,height=540,spp=16)
sc=R.Scene(p); ren=R.Renderer(sc); ren.set_counters(True)
cam=sc.camera(0)
img,st=ren.render(cam); a=st.as_dict(); print('with NEE', a)
cam2=sc.camera(0,next_event_estimation=0)
img,st2=ren.render(cam2); b=st2.as_dict(); print('no NEE  ', b)
cl_rays=b['primary_rays']+b['secondary_rays']
print('closest: rays',cl_rays,'nodes/ray',b['nodes_visited']/cl_rays,'tri/ray',b['triangle_tests']/cl_rays)
sh=a['shadow_rays']; print('anyhit: rays',sh,'nodes/ray',(a['nodes_visited']-b['nodes_visited'])/sh,'tri/ray',(a['triangle_tests']-b['triangle_tests'])/sh)
