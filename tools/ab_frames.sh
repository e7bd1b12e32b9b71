# A/B of build variants of the GPU library on the GPU box by whole-frame time (tools/frame_time.py), each variant twice,
# alternating: tools/ab_frames.sh [spp]   (variants/librt_gpu_<n>.so built beforehand by tools/build_variants.sh)
cd $GRAFT_REPO_ROOT
cp raytracer_amd/librt_gpu.so /tmp/librt_gpu_orig.so
for round in 1 2; do
  for v in variants/librt_gpu_*.so; do
    echo "== $v $(cat ${v%.so}.flags)"
    cp $v raytracer_amd/librt_gpu.so
    timeout -k 10 200 python3 -u tools/frame_time.py ${1:-64} || exit 1
  done
done
cp /tmp/librt_gpu_orig.so raytracer_amd/librt_gpu.so
