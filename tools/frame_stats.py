"""One timed 4K / 64 spp frame of the 1M-triangle scene: per-kernel event times and the frame's ray statistics."""
import sys
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = scenegen.write_path_traced_scene('/tmp/scn_kt', width=3840, height=2160, spp=spp)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
ren.set_profiling(True)
img, st = ren.render(cam)
kt = ren.kernel_times()
print("frame kernel_ms %.1f" % st.kernel_ms, {k: round(v[1], 1) for k, v in kt.items()})
print("primary %d secondary %d shaded %d shadow(evaluated samples) %d shadow traced %d  image mean %.6f" %
      (st.primary_rays, st.secondary_rays, st.shaded_hits, st.shadow_rays, st.shadow_rays_traced, float(img.mean())))
