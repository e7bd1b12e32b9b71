# Runs a command with variants/librt_gpu_<n>.so in place of the library (on the GPU box's scratch copy): tools/run_variant.sh <n> <command...>
cd $GRAFT_REPO_ROOT
cp raytracer_amd/librt_gpu.so /tmp/librt_gpu_orig.so && cp variants/librt_gpu_$1.so raytracer_amd/librt_gpu.so
shift
"$@"; rc=$?
cp /tmp/librt_gpu_orig.so raytracer_amd/librt_gpu.so
exit $rc
