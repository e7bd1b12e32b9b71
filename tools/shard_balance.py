"""Device time of each rank's shard of one frame (all shards rendered one after another on this GPU): how evenly the
tile -> rank map spreads the work.  usage: python tools/shard_balance.py [world] [spp]"""
import sys
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p = scenegen.write_path_traced_scene('/tmp/scn_sb', width=3840, height=2160, spp=spp)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam, rank=0, world=world)
ms = []
for r in range(world):
    img, st = ren.render(cam, rank=r, world=world)
    ms.append(st.kernel_ms)
img, st = ren.render(cam)
print('world', world, 'shard ms', [round(m, 1) for m in ms], 'max/mean %.3f' % (max(ms) / (sum(ms) / world)), 'whole frame %.1f ms -> ideal shard %.1f ms, slowest shard %.1f ms (efficiency %.2f)' % (st.kernel_ms, st.kernel_ms / world, max(ms), st.kernel_ms / world / max(ms)))
