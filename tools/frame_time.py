"""Frame time of a 4K path-traced frame of the 1M-triangle scene without per-kernel events (three frames, each printed),
then the per-kernel event times of a fourth; environment settings pass through: RT_PAIR_GATHER=0 python tools/frame_time.py [spp]"""
import hashlib, sys
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = scenegen.write_path_traced_scene('/tmp/scn_kt', width=3840, height=2160, spp=spp)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
ms = []
for k in range(3):
    img, st = ren.render(cam)
    ms.append(round(st.kernel_ms, 1))
print('frame ms', ms, 'image sha', hashlib.sha256(img.tobytes()).hexdigest()[:12], 'sum', float(img.sum(dtype='float64')))
ren.set_profiling(True)
img, st = ren.render(cam)
print('with events: frame %.1f' % st.kernel_ms, {k: round(v[1], 1) for k, v in ren.kernel_times().items()})
