# A/B of environment settings on the GPU box (per-kernel times of a 4K / 64 spp frame of the 1M-triangle scene):
#   tools/ab_env.sh "" "RT_TILE_ORDER=0" ...
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  echo "== ${e:-default}"
  env $e timeout -k 10 200 python3 -u -c '
import sys
sys.path.insert(0, ".")
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene("/tmp/scn_kt", width=3840, height=2160, spp=64)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
ren.set_profiling(True)
img, st = ren.render(cam)
kt = ren.kernel_times()
print("frame kernel_ms %.1f" % st.kernel_ms, {k: round(v[1], 1) for k, v in kt.items()})
' || exit 1
done
