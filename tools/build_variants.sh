# Builds variants of the GPU library with extra compiler flags, one per argument, into variants/ (shipped to the GPU box,
# not committed): tools/build_variants.sh "-DRT_TRACE_WAVES=6" "-DRT_TRACE_WAVES=8 -DRT_LDS_STACK=10"
mkdir -p variants && rm -f variants/*
n=0
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-vector-combine -fPIC -std=c++17 -Iinclude -Iraytracer_amd/csrc/gpu $flags -c raytracer_amd/csrc/gpu/render.hip -o variants/render_$n.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/render_$n.o build/bvh_gpu.o -o variants/librt_gpu_$n.so || exit 1   # (make all builds build/bvh_gpu.o)
  rm -f variants/render_$n.o
  echo "$flags" > variants/librt_gpu_$n.flags
  n=$((n+1))
done
