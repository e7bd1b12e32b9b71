cd $GRAFT_REPO_ROOT
cp raytracer_amd/librt_gpu.so /tmp/orig.so && cp variants/librt_gpu_0.so raytracer_amd/librt_gpu.so && timeout -k 10 200 python3 -u tools/trace_profile_c3.py; rc=$?; cp /tmp/orig.so raytracer_amd/librt_gpu.so; exit $rc
