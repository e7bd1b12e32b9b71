"""Diagnostic (library built with -DRT_TRACE_PROFILE=3 for the closest-hit kernel, =4 for the any-hit kernel): how long the
trace kernel's waves live compared with their launches, and how much of that is after their shard's queue ran dry."""
import os, sys
os.environ.setdefault('RT_WIDE_NODES', '1')
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene('/tmp/scn_kt', width=3840, height=2160, spp=64)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
ren.set_profiling(True)
img, st = ren.render(cam)
kt = ren.kernel_times()
d = st.as_dict()
life, dry, waves, its = d['nodes_visited'], d['triangle_tests'], d['sphere_tests'], d['plane_tests']
which = 'trace_any' if len(sys.argv) > 1 and sys.argv[1] == 'any' else 'trace_closest'
launches, ms = kt[which]
print(which, 'launches', launches, 'total %.1f ms' % ms, 'waves per launch %.0f' % (waves / launches))
print('mean wave lifetime %.3f ms of a mean launch of %.3f ms (%.1f %%)' % (life / waves / 1e5, ms / launches, 100 * (life / waves / 1e5) / (ms / launches)))
print('of a wave\'s lifetime after its shard\'s queue ran dry: %.1f %%' % (100.0 * dry / life), ' iterations per wave and launch %.0f' % (its / waves))
