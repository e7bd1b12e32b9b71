"""Config 3 at full size: the `distributed` test scene (area lights, glossy mirror, depth of field, motion blur) at
1920x1080 with its 16 samples per pixel -- every light sample has its own shadow ray and record here (no sphere emitter),
so this is the record path of the lighting kernels with several batches.  Prints frame times and checks that two
renders and a two-shard render agree."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import raytracer_amd as R
from golden import scenes as GS

sc = R.Scene(GS.distributed_scene("/tmp/c3_scene"))
cam = sc.camera(0, resx=1920, resy=1080)
ren = R.Renderer(sc)
a, st = ren.render(cam)
b, st = ren.render(cam)
print("frame %.1f ms, %.0f Mrays/s, rays %d (shadow %d, traced %d)" % (st.kernel_ms, st.total_rays / st.kernel_ms / 1e3, st.total_rays, st.shadow_rays, st.shadow_rays_traced))
err = np.abs(a - b) / np.maximum(np.abs(b), 1 / 255)
print("run-to-run max rel err %.2e" % err.max())
assert err.max() <= 1e-5
total = np.zeros_like(a)
for rank in range(2):
    part = np.zeros_like(a)
    ren.render(cam, out=part, rank=rank, world=2)
    total += part
err = np.abs(total - a) / np.maximum(np.abs(a), 1 / 255)
print("two shards vs one: max rel err %.2e" % err.max())
assert err.max() <= 1e-5
ren.set_counters(True)
c, sc_ = ren.render(cam)
err = np.abs(c - a) / np.maximum(np.abs(a), 1 / 255)
print("counted frame %.1f ms, max rel err %.2e, shadow rays %d" % (sc_.kernel_ms, err.max(), sc_.shadow_rays))
assert err.max() <= 1e-5 and sc_.shadow_rays == st.shadow_rays
