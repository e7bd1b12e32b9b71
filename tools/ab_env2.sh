# like ab_env.sh, but without forcing one stream: tools/ab_env2.sh "" "RT_ONE_STREAM=1" ...
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
for k in range(2):
    img, st = ren.render(cam)
    print("frame kernel_ms %.1f total_ms %.1f" % (st.kernel_ms, st.total_ms))
' || exit 1
done
