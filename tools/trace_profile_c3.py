"""Diagnostic (library built with -DRT_TRACE_PROFILE=2): the any-hit trace kernel's wave iterations on the textured
1080p / 64 spp frame (config 3): how they split into node / leaf / pop steps and how many lanes take part in each."""
import os, sys
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
import bench
w = bench.WORKLOADS['textured_1080p_64spp']
p = bench.make_scene(scenegen, '/tmp/scn_tp_c3', w, 'textured_1080p_64spp')
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
img, st = ren.render(cam)
d = st.as_dict()
it = d['any_hit_sphere_tests']
print('wave iterations', it, 'busy lanes per iteration %.1f' % (d['any_hit_plane_tests'] / it))
for name, a, b in (('node', 'nodes_visited', 'triangle_tests'), ('leaf', 'sphere_tests', 'plane_tests'), ('pop', 'any_hit_nodes_visited', 'any_hit_triangle_tests')):
    print(f"{name}: in {d[a] / it:.3f} of the iterations, {d[b] / max(1, d[a]):.1f} lanes each")
rays = d['shadow_rays_traced']
print('per ray: node steps %.2f leaf steps %.2f pops %.2f; rays %d' % (d['triangle_tests'] / rays, d['plane_tests'] / rays, d['any_hit_triangle_tests'] / rays, rays))
