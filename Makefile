# Builds the product libraries in-tree and the CPU oracle used by the tests.
#   raytracer_amd/librt_gpu.so   HIP rendering core (gfx950), C ABI include/gpu_render.h
#   raytracer_amd/librt_host.so  host library: scene ingest, BVH build, tone map, image files
#   raytracer                    command-line front end (same argv contract as the reference)
#   oracle/liboracle.so          CPU checker (tests only); oracle/_ref/* when the reference is present
HIPCC ?= hipcc
CXX ?= g++
HOST_SRCS = $(addprefix raytracer_amd/csrc/host/,ingest.cpp ply.cpp bvh_build.cpp image_io.cpp jpeg.cpp tonemap.cpp scene_cache.cpp host_api.cpp)
HOST_HDRS = $(wildcard raytracer_amd/csrc/host/*.h) include/gpu_render.h include/rt_host.h
HIPFLAGS = --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-vector-combine -fPIC -std=c++17 -Iinclude -Iraytracer_amd/csrc/gpu

all: raytracer_amd/librt_gpu.so raytracer_amd/librt_host.so raytracer oracle tests/tools/libfake_rccl.so tools/calibrate_fetch

# two translation units: the render path, and the device BVH builder (it pulls in the radix sort of hipCUB, slow to compile)
GPU_HDRS = $(wildcard raytracer_amd/csrc/gpu/*.h) include/gpu_render.h
build/render.o: raytracer_amd/csrc/gpu/render.hip $(GPU_HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
build/bvh_gpu.o: raytracer_amd/csrc/gpu/bvh_gpu.hip raytracer_amd/csrc/gpu/rt_bvh_gpu.h raytracer_amd/csrc/gpu/rt_device.h raytracer_amd/csrc/gpu/rt_devmem.h raytracer_amd/csrc/gpu/rt_math.h include/gpu_render.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
raytracer_amd/librt_gpu.so: build/render.o build/bvh_gpu.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $^ -o $@

# (the scene cache keys its files by a hash of the host sources: a cache written by another ingest or tree builder is not used)
raytracer_amd/librt_host.so: $(HOST_SRCS) $(HOST_HDRS)
	$(CXX) -std=c++17 -O2 -fPIC -shared -Wall -Iinclude -Iraytracer_amd/csrc/host -DRT_HOST_SOURCE_STAMP=\"$$(cat $(HOST_SRCS) $(HOST_HDRS) | sha256sum | cut -c1-32)\" $(HOST_SRCS) -lz -pthread -o $@

raytracer: raytracer_amd/csrc/host/main.cpp raytracer_amd/librt_host.so raytracer_amd/librt_gpu.so
	$(CXX) -std=c++17 -O2 -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Iraytracer_amd/csrc/host $< -L/opt/rocm/lib -lamdhip64 -Lraytracer_amd -lrt_host -lrt_gpu -Wl,-rpath,'$$ORIGIN/raytracer_amd' -pthread -o $@

# measurement: independent 64-byte gathers at known byte counts (bench.py times it for the gather ceiling of its roofline)
tools/calibrate_fetch: tools/calibrate_fetch.hip
	$(HIPCC) --offload-arch=gfx950 -O3 $< -o $@

# test infrastructure: a stand-in for librccl.so that carries the exchange between threads of one process (RT_RCCL_LIB)
tests/tools/libfake_rccl.so: tests/tools/fake_rccl.cpp
	$(CXX) -std=c++17 -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $< -L/opt/rocm/lib -lamdhip64 -pthread -o $@

oracle: raytracer_amd/librt_gpu.so
	$(MAKE) -C oracle all

clean:
	rm -f raytracer_amd/*.so raytracer oracle/liboracle.so
.PHONY: all oracle clean
