"""Wall time per frame of the 70K-triangle 1080p / 1 spp scene rendered repeatedly into a device buffer (no profiling
events): RT_FRAME_GRAPH=1 (default) replays the frame's launches from a HIP graph, =0 issues them one by one."""
import sys, time
sys.path.insert(0, '.')
import torch
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_mesh_scene('/tmp/scn_rep', name='rep', n=188, width=1920, height=1080)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
fb = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.float32, device='cuda')
for _ in range(20):
    ren.render(cam, out=fb.data_ptr())
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    _, st = ren.render(cam, out=fb.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
print('%.3f ms per frame (device time of the last frame %.3f ms)' % (dt, st.kernel_ms))
