"""Diagnostic (library built with -DRT_TRACE_PROFILE): how the closest-hit trace kernel's wave iterations split into
node / leaf / pop steps and how many lanes take part in each."""
import os, sys
os.environ.setdefault('RT_WIDE_NODES', '1')
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene('/tmp/scn_tp', width=1920, height=1080, spp=16)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
img, st = ren.render(cam)
d = st.as_dict()
it = d['any_hit_sphere_tests']
print('wave iterations', it, 'busy lanes per iteration %.1f' % (d['any_hit_plane_tests'] / it))
for name, a, b in (('node', 'nodes_visited', 'triangle_tests'), ('leaf', 'sphere_tests', 'plane_tests'), ('pop', 'any_hit_nodes_visited', 'any_hit_triangle_tests')):
    print(f"{name}: in {d[a] / it:.3f} of the iterations, {d[b] / max(1, d[a]):.1f} lanes each")
rays = d['shadow_rays_traced'] if len(sys.argv) > 1 and sys.argv[1] == 'any' else d['primary_rays'] + d['secondary_rays']
print('per ray: node steps %.2f leaf steps %.2f pops %.2f' % (d['triangle_tests'] / rays, d['plane_tests'] / rays, d['any_hit_triangle_tests'] / rays))
