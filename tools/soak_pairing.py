"""Soak of the paired schedule (gather of a generation beside the next generation's shading): the same 4K path-traced frame
many times over, each compared with a frame rendered in turn.  A race between the two streams would show as a difference
beyond the order of a sample's float additions.  usage: python tools/soak_pairing.py [frames] [spp]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as R
from raytracer_amd import scenegen
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
p = scenegen.write_path_traced_scene('/tmp/scn_soak', width=3840, height=2160, spp=spp)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
os.environ["RT_PAIR_GATHER"] = "0"
ref, st0 = ren.render(cam)
os.environ["RT_PAIR_GATHER"] = "2"
worst = 0.0
for k in range(frames):
    img, st = ren.render(cam)
    assert (st.primary_rays, st.secondary_rays, st.shadow_rays, st.shadow_rays_traced) == (st0.primary_rays, st0.secondary_rays, st0.shadow_rays, st0.shadow_rays_traced), k
    err = float((np.abs(img - ref) / np.maximum(np.abs(ref), 1.0 / 255.0)).max())
    worst = max(worst, err)
    assert err < 1e-5, (k, err)
print(f"{frames} paired frames of 3840x2160x{spp}: largest relative difference from the frame rendered in turn {worst:.2e}")
