# A/B of build variants of the GPU library on the GPU box, textured 1080p / 64 spp frame (config 3); variants as in
# tools/ab_variants.sh (built beforehand by tools/build_variants.sh into variants/).
cd $GRAFT_REPO_ROOT
cp raytracer_amd/librt_gpu.so /tmp/librt_gpu_orig.so
for v in variants/librt_gpu_*.so; do
  echo "== $v $(cat ${v%.so}.flags)"
  cp $v raytracer_amd/librt_gpu.so
  timeout -k 10 200 python3 -u bench.py --timed-frames-only --workload textured_1080p_64spp --steps 2 --warmup 1 || exit 1
done
cp /tmp/librt_gpu_orig.so raytracer_amd/librt_gpu.so
