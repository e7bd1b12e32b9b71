"""Per-kernel device time of one rank's shard against the whole frame's (same GPU): where a shard loses against
frame / world.  usage: python tools/shard_kernels.py [world] [spp]"""
import sys
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
p = scenegen.write_path_traced_scene('/tmp/scn_sb', width=3840, height=2160, spp=spp)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam, rank=0, world=world)
ren.set_profiling(True)
img, st = ren.render(cam, rank=3 % world, world=world)
shard = {k: v for k, v in ren.kernel_times().items()}
ren.set_profiling(True)
img, st2 = ren.render(cam)
whole = ren.kernel_times()
print('shard %.1f ms, frame %.1f ms, frame/world %.1f' % (st.kernel_ms, st2.kernel_ms, st2.kernel_ms / world))
for k in whole:
    print(f"  {k:14s} shard {shard[k][0]:4d} launches {shard[k][1]:8.2f} ms | frame {whole[k][0]:4d} launches {whole[k][1]:8.2f} ms, /world {whole[k][1] / world:8.2f}  ratio {shard[k][1] / (whole[k][1] / world):.3f}")
