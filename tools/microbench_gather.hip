// Micro-benchmark: how fast can the 64 lanes of a wave each fetch their own 64-byte record (a BVH node) from a table?
//   mode 0  every lane reads its record with four 16-byte loads (what the trace kernel does)
//   mode 1  cooperative: the four lanes of a quad read the four 16-byte parts of ONE record with one load instruction
//           (four instructions cover the wave's 64 records), the parts go through LDS to the lane that owns the record
//   mode 2  every lane reads only the first 16 bytes of its record (one load instruction; cost of a single divergent load)
//   mode 3  cooperative as mode 1, but the loads write LDS directly (global_load_lds_dwordx4: no register staging, no
//           ds_write); the four passes land 1040 bytes apart so that the lanes' 16-byte reads spread over the banks
// `active` of 64 lanes take part (the others idle, as in a partly filled traversal wave).
// All indices are masked to the table size; nothing is written to global memory except one sum per thread.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o tools/microbench_gather
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) gather_kernel(const float4 *__restrict__ table, unsigned mask, int iters, int active, float *__restrict__ out) {
    __shared__ float4 stage[4][260];   // per wave: 64 records of 4 parts (mode 3: four passes of 1040 bytes)
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const bool on = (int)lane < active;   // the first `active` lanes (whole quads when active is a multiple of 4)
    float acc = 0.f;
    unsigned idx = hash32(gid) & mask;
    for (int it = 0; it < iters; ++it) {
        float4 a = make_float4(0, 0, 0, 0), b = a, c = a, d = a;
        if (MODE == 0) {
            if (on) {
                const float4 *r = table + 4 * (size_t)idx;
                a = r[0], b = r[1], c = r[2], d = r[3];
            }
        } else if (MODE == 2) {
            if (on) a = table[4 * (size_t)idx];
        } else if (MODE == 3) {
            char *base = (char *)&stage[wave][0];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // pass p: every quad fetches the record of its lane p, 16 bytes per lane
                const int src = (int)(lane & ~3u) + p;
                const unsigned sidx = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)idx) & mask;
                if (src < active)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(table + 4 * (size_t)sidx + (lane & 3u)),
                                                     (__attribute__((address_space(3))) void *)(base + 1040 * p), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (on) {
                const float4 *r = (const float4 *)(base + 1040 * (lane & 3u) + 64 * (lane >> 2));
                a = r[0], b = r[1], c = r[2], d = r[3];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next pass overwrites the stage
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int src = 16 * p + (int)(lane >> 2);          // the lane whose record this quad fetches in pass p
                const unsigned sidx = (unsigned)__shfl((int)idx, src, 64) & mask;   // every lane holds a valid idx
                if (src < active) stage[wave][64 * p + lane] = table[4 * (size_t)sidx + (lane & 3u)];
            }
            __builtin_amdgcn_wave_barrier();
            if (on) {
                const float4 *r = &stage[wave][4 * lane];
                a = r[0], b = r[1], c = r[2], d = r[3];
            }
            __builtin_amdgcn_wave_barrier();
        }
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);   // all 64 bytes are used
        // next record: depends on the data just read, as a traversal's next node does
        // (keyed by the thread as well: with one function for all threads their walks would merge into a few records)
        idx = hash32(idx + (unsigned)it * 0x9e3779b9u + gid * 0x85ebca6bu + (__float_as_uint(a.x) & 1u)) & mask;
    }
    out[gid] = acc;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * (argc > 2 ? atoi(argv[2]) : 6);   // 6 waves per SIMD, as the trace kernel
    float *out;
    CHECK(hipMalloc((void **)&out, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned sizes[3] = {1u << 15, 1u << 19, 1u << 21};   // records: 2 MiB (in every L2), 32 MiB, 128 MiB
    for (unsigned n : sizes) {
        std::vector<float4> host((size_t)n * 4);
        for (size_t i = 0; i < host.size(); ++i) host[i] = make_float4((float)(i & 255), 1.f, 2.f, 3.f);
        float4 *table;
        CHECK(hipMalloc((void **)&table, host.size() * sizeof(float4)));
        CHECK(hipMemcpy(table, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice));
        const int actives[3] = {64, 44, 20};
        for (int active : actives) {
            for (int mode = 0; mode < 4; ++mode) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0, 0));
                    if (mode == 0)
                        hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, table, n - 1, iters, active, out);
                    else if (mode == 1)
                        hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, n - 1, iters, active, out);
                    else if (mode == 2)
                        hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, table, n - 1, iters, active, out);
                    else
                        hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(256), 0, 0, table, n - 1, iters, active, out);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipGetLastError());
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double records = (double)blocks * 4 * active * iters;
                printf("table %4u KiB  active %2d  mode %d  %8.3f ms  %7.2f G records/s  %6.1f ns per iteration\n", n / 16, active, mode, best,
                       records / best / 1e6, best * 1e6 / iters);
                fflush(stdout);
            }
        }
        CHECK(hipFree(table));
    }
    return 0;
}
