// Calibration of rocprofv3's FETCH_SIZE for the access pattern of the trace kernel: every lane fetches its own 64-byte
// record (four 16-byte loads) from a table at a random index.  Each kernel below requests a known number of bytes from
// a table far larger than the caches, so nearly every record is an L2 miss; FETCH_SIZE of the same launch divided into
// that number is the factor to apply to FETCH_SIZE of the trace kernel.
//   stream          16 bytes per lane, consecutive                (the pattern the x2 rule of gfx950 was measured on)
//   gather64        one 64-byte record per lane, records packed   (neighbouring records share a 128-byte line)
//   gather64_s128   one 64-byte record per lane at 128-byte stride (the other half of every line is never wanted)
//   gather16        the first 16 bytes of a random 64-byte record
//   gather128       one 128-byte record (a whole line) per lane
// Prints one line per launch: kernel name, bytes requested, milliseconds (HIP events).
//   hipcc --offload-arch=gfx950 -O3 tools/calibrate_fetch.hip -o tools/calibrate_fetch
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                    \
        }                                                               \
    } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) stream(const float4 *__restrict__ table, size_t n16, float *__restrict__ out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += table[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// PARTS 16-byte loads per record, records STRIDE16 float4 apart; `mask` + 1 records in the table
template <int PARTS, int STRIDE16>
__device__ __forceinline__ float gather_body(const float4 *__restrict__ table, unsigned mask, int iters) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    float acc = 0.f;
    unsigned idx = hash32(gid);
    for (int it = 0; it < iters; ++it) {
        const float4 *r = table + (size_t)STRIDE16 * (idx & mask);
#pragma unroll
        for (int p = 0; p < PARTS; ++p) acc += r[p].x;
        idx = hash32(idx + (unsigned)it);
    }
    return acc;
}
__global__ void __launch_bounds__(256) gather64(const float4 *__restrict__ t, unsigned mask, int iters, float *__restrict__ out) {
    out[blockIdx.x * 256 + threadIdx.x] = gather_body<4, 4>(t, mask, iters);
}
__global__ void __launch_bounds__(256) gather64_s128(const float4 *__restrict__ t, unsigned mask, int iters, float *__restrict__ out) {
    out[blockIdx.x * 256 + threadIdx.x] = gather_body<4, 8>(t, mask, iters);
}
__global__ void __launch_bounds__(256) gather16(const float4 *__restrict__ t, unsigned mask, int iters, float *__restrict__ out) {
    out[blockIdx.x * 256 + threadIdx.x] = gather_body<1, 4>(t, mask, iters);
}
__global__ void __launch_bounds__(256) gather128(const float4 *__restrict__ t, unsigned mask, int iters, float *__restrict__ out) {
    out[blockIdx.x * 256 + threadIdx.x] = gather_body<8, 8>(t, mask, iters);
}

int main(int argc, char **argv) {
    const size_t table_bytes = (argc > 1 ? (size_t)atol(argv[1]) : 2048) << 20;   // MiB, a power of two
    const int iters = argc > 2 ? atoi(argv[2]) : 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8;
    float4 *table;
    float *out;
    CHECK(hipMalloc((void **)&table, table_bytes));
    CHECK(hipMemset(table, 0, table_bytes));
    CHECK(hipMalloc((void **)&out, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double lanes = (double)blocks * 256;
    auto timed = [&](const char *name, double bytes, auto launch) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            launch();
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s table_MiB %zu requested_bytes %.0f ms %.4f GBps %.1f\n", name, table_bytes >> 20, bytes, ms, bytes / ms / 1e6);
        }
    };
    timed("stream", (double)table_bytes, [&] { hipLaunchKernelGGL(stream, dim3(blocks), dim3(256), 0, 0, table, table_bytes / 16, out); });
    timed("gather64", lanes * iters * 64, [&] { hipLaunchKernelGGL(gather64, dim3(blocks), dim3(256), 0, 0, table, (unsigned)(table_bytes / 64 - 1), iters, out); });
    timed("gather64_s128", lanes * iters * 64, [&] { hipLaunchKernelGGL(gather64_s128, dim3(blocks), dim3(256), 0, 0, table, (unsigned)(table_bytes / 128 - 1), iters, out); });
    timed("gather16", lanes * iters * 16, [&] { hipLaunchKernelGGL(gather16, dim3(blocks), dim3(256), 0, 0, table, (unsigned)(table_bytes / 64 - 1), iters, out); });
    timed("gather128", lanes * iters * 128, [&] { hipLaunchKernelGGL(gather128, dim3(blocks), dim3(256), 0, 0, table, (unsigned)(table_bytes / 128 - 1), iters, out); });
    return 0;
}
