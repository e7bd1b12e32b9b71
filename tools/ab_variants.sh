# A/B of build variants of the GPU library on the GPU box: tools/ab_variants.sh "<flags A>" "<flags B>" ...
# Each variant is built in the CPU container beforehand as gpurun_out/variants/librt_gpu_<n>.so by tools/build_variants.sh;
# here each one is copied over raytracer_amd/librt_gpu.so in the box's scratch copy and timed (per-kernel event times).
cd $GRAFT_REPO_ROOT
cp raytracer_amd/librt_gpu.so /tmp/librt_gpu_orig.so
for v in variants/librt_gpu_*.so; do
  echo "== $v $(cat ${v%.so}.flags)"
  cp $v raytracer_amd/librt_gpu.so
  timeout -k 10 200 python3 -u -c '
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
import hashlib
print("frame kernel_ms %.1f" % st.kernel_ms, {k: round(v[1], 1) for k, v in kt.items()}, "image sha", hashlib.sha256(img.tobytes()).hexdigest()[:12])
' || exit 1
done
cp /tmp/librt_gpu_orig.so raytracer_amd/librt_gpu.so
