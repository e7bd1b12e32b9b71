// Tone mapping and 8-bit quantisation of the HDR framebuffer on the device: apply_tonemap and the plain LDR
// conversion of the reference (src/main.cpp:169-345, :579-583), so that a frame that is only wanted as PNG never
// leaves HBM as floats.  Included by render.hip (shares its error string and HIP_OK).
//
// Steps: luminance + log-average (per-thread double sums, fixed-order tree per block, fixed-order sum of the block
// results: the same value every run), exposure, the burn-percent white point as an exact order statistic (radix
// select over the float bit patterns, four 8-bit passes), then the per-pixel curve, saturation and gamma.
// Everything between the first launch and the last stays on the device: the block sums are added and turned into the
// exposure by a one-wave kernel, the select state lives in device memory, and the only scratch -- 1024 partial sums, the
// select state and a 256-bin histogram -- is a fixed array inside the library, so a call costs eleven short launches and
// no allocation, copy or wait in between (a 3840x2160 frame: see tools/bench_tonemap.py).
#pragma once

namespace {

constexpr int TM_THREADS = 256;
constexpr int TM_BLOCKS = 1024;

__device__ __forceinline__ float tm_luminance(const float *p) { return 0.2126f * p[0] + 0.7152f * p[1] + 0.0722f * p[2]; }

// lum[i] and one double partial sum of log(1e-4 + max(lum, 0)) per block.  With `lum` null no luminance plane is kept:
// the later kernels then evaluate tm_luminance again from `hdr` (the same instructions, so the same bits).
__global__ void __launch_bounds__(TM_THREADS) tm_luminance_kernel(const float *__restrict__ hdr, size_t n, float *__restrict__ lum,
                                                                   double *__restrict__ partial) {
    __shared__ double sums[TM_THREADS];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * TM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * TM_THREADS) {
        const float l = tm_luminance(hdr + 3 * i);
        if (lum) lum[i] = l;
        acc += (double)logf(1e-4f + (l < 0.0f ? 0.0f : l));
    }
    sums[threadIdx.x] = acc;
    __syncthreads();
    for (int half = TM_THREADS / 2; half > 0; half >>= 1) {
        if ((int)threadIdx.x < half) sums[threadIdx.x] += sums[threadIdx.x + half];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sums[0];
}

// state[0] = bit-pattern prefix found so far, state[1] = rank still to find inside it, state[2] = result bits
// state[3] = the exposure (float bits), set by tm_prepare_kernel
__global__ void __launch_bounds__(TM_THREADS) tm_histogram_kernel(const float *__restrict__ hdr, const float *__restrict__ lum, size_t n, int pass,
                                                                   const unsigned *__restrict__ state, unsigned *__restrict__ hist) {
    __shared__ unsigned local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
    const float exposure = __uint_as_float(state[3]);
    const int shift = 24 - 8 * pass;
    const unsigned prefix = state[0];
    const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (size_t i = (size_t)blockIdx.x * TM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * TM_THREADS) {
        const float l = lum ? lum[i] : tm_luminance(hdr + 3 * i);
        const unsigned bits = __float_as_uint((l < 0.0f ? 0.0f : l) * exposure);   // non-negative: bit order = value order
        if ((bits & mask) == prefix) atomicAdd(&local[(bits >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (local[threadIdx.x]) atomicAdd(&hist[threadIdx.x], local[threadIdx.x]);
}

// One workgroup of 256: the bin in which the wanted rank falls (bins in ascending order), the rank inside it, and the
// histogram cleared for the next pass.
__global__ void __launch_bounds__(256) tm_select_kernel(int pass, unsigned *__restrict__ state, unsigned *__restrict__ hist) {
    __shared__ unsigned incl[256];
    const unsigned b = threadIdx.x, c = hist[b];
    incl[b] = c;
    __syncthreads();
    for (unsigned step = 1; step < 256; step <<= 1) {   // inclusive prefix sums
        const unsigned add = b >= step ? incl[b - step] : 0;
        __syncthreads();
        incl[b] += add;
        __syncthreads();
    }
    const int shift = 24 - 8 * pass;
    const unsigned k = state[1], prefix_so_far = state[0];
    hist[b] = 0;
    __syncthreads();   // (every thread has read the state)
    const unsigned below = incl[b] - c;
    // below <= k < below + c for exactly one bin; a rank beyond the total (not reached: the rank is below n) takes the last
    if ((c > 0 && k >= below && k < incl[b]) || (b == 255 && k >= incl[255])) {
        const unsigned prefix = prefix_so_far | b << shift;
        state[0] = prefix;
        state[1] = k - below;
        if (pass == 3) state[2] = prefix;
    }
}

// One wave: the block sums in a fixed order -> log-average luminance -> exposure (src/main.cpp:186-200), and the select
// state for a frame of n pixels: state = {0, rank wanted, 0, exposure bits}, histogram cleared.
__global__ void __launch_bounds__(64) tm_prepare_kernel(const double *__restrict__ partial, int blocks, double n, float key, unsigned rank, unsigned *__restrict__ state,
                                                         unsigned *__restrict__ hist) {
    __shared__ double sums[64];
    double acc = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 64) acc += partial[b];
    sums[threadIdx.x] = acc;
    __syncthreads();
    for (int half = 32; half > 0; half >>= 1) {
        if ((int)threadIdx.x < half) sums[threadIdx.x] += sums[threadIdx.x + half];
        __syncthreads();
    }
    for (int b = threadIdx.x; b < 256; b += 64) hist[b] = 0;
    if (threadIdx.x == 0) {
        const float avg = (float)exp(sums[0] / n);
        state[0] = 0;
        state[1] = rank;
        state[2] = 0;
        state[3] = __float_as_uint(avg > 1e-6f ? key / avg : 1.0f);
    }
}

struct TonemapParams {
    int tmo;
    float saturation, inv_gamma;
    int use_state_white;   // white point = max(value with bits state[2], 1e-6), else 1e9
};

__device__ __forceinline__ float tm_clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
__device__ __forceinline__ float tm_aces(float x) { return (x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f); }
__device__ __forceinline__ float tm_filmic(float x) {
    const float A = 0.22f, B = 0.30f, C = 0.10f, D = 0.20f, E = 0.01f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - (E / F);
}

__global__ void __launch_bounds__(TM_THREADS) tm_apply_kernel(const float *__restrict__ hdr, const float *__restrict__ lum, size_t n, TonemapParams p,
                                                               const unsigned *__restrict__ state, uint8_t *__restrict__ out) {
    float white = 1e9f;
    if (p.use_state_white) {
        white = __uint_as_float(state[2]);
        if (!(white > 1e-6f)) white = 1e-6f;
    }
    const float white_sq = white * white;
    const float exposure = __uint_as_float(state[3]);
    float denom = 1.0f;
    if (p.tmo == RT_TMO_ACES) denom = fmaxf(tm_aces(white), 1e-6f);
    if (p.tmo == RT_TMO_FILMIC) denom = fmaxf(tm_filmic(white), 1e-6f);
    for (size_t i = (size_t)blockIdx.x * TM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * TM_THREADS) {
        const float y_in = lum ? lum[i] : tm_luminance(hdr + 3 * i);
        const float L = (y_in < 0.0f ? 0.0f : y_in) * exposure;
        float y_out = 0.0f;
        if (p.tmo == RT_TMO_PHOTOGRAPHIC)
            y_out = (L * (1.0f + (L / white_sq))) / (1.0f + L);
        else if (p.tmo == RT_TMO_ACES)
            y_out = tm_aces(L) / denom;
        else if (p.tmo == RT_TMO_FILMIC)
            y_out = tm_filmic(L) / denom;
        y_out = tm_clamp01(y_out);
        for (int k = 0; k < 3; ++k) {
            float c = 0.0f;
            if (!(y_in < 1e-6f)) {
                float ratio = hdr[3 * i + k] / y_in;
                if (fabsf(p.saturation - 1.0f) > 1e-4f) ratio = powf(ratio < 0.0f ? 0.0f : ratio, p.saturation);
                c = y_out * ratio;
            }
            out[3 * i + k] = (uint8_t)lroundf(255.0f * powf(tm_clamp01(c), p.inv_gamma));
        }
    }
}

__global__ void __launch_bounds__(TM_THREADS) tm_quantize_kernel(const float *__restrict__ hdr, size_t n, uint8_t *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * TM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * TM_THREADS) {
        float v = hdr[i] * 255.0f;
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);   // std::clamp(v, 0, 255); a NaN stays and converts to 0
        out[i] = (uint8_t)v;
    }
}

// `device_out`: the GPU that holds the memory (when it is device memory)
bool tm_on_device(const void *p, int *device_out = nullptr) {
    hipPointerAttribute_t attr;
    const bool dev = hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (dev && device_out) *device_out = attr.device;
    return dev;
}

// Device copies of caller buffers that live in host memory; released on scope exit.
struct TmBuffers {
    std::vector<void *> owned;
    ~TmBuffers() {
        for (void *p : owned) (void)hipFree(p);
    }
    void *get(size_t bytes) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        owned.push_back(p);
        return p;
    }
};

// ---- the tone mapper's scratch -----------------------------------------------------------------------------------
// TM_BLOCKS double partial sums, then the radix select's state (4 words) and histogram (256 words): all the device memory
// a tone map needs beside its input and output.  One copy per GPU for the life of the process, inside the library, so
// that no call allocates: rt_tonemap_noalloc promises that, and rt_tonemap saves five allocations per frame by it.  Calls on
// one GPU take that GPU's lock for their launches (the array is one per GPU).
constexpr size_t TM_SCRATCH_PARTIAL_BYTES = TM_BLOCKS * sizeof(double);
constexpr size_t TM_SCRATCH_BYTES = TM_SCRATCH_PARTIAL_BYTES + (4 + 256) * sizeof(unsigned);
static __device__ __align__(256) unsigned char scratch[(TM_SCRATCH_BYTES + 255) / 256 * 256];

// The device address of `scratch` on the current device.  In host code the array's name is only a handle for
// hipGetSymbolAddress; every GPU holds its own copy at its own address, looked up once per device.
constexpr int TM_MAX_DEVICES = 64;
std::mutex g_tm_scratch_mutex;
void *g_tm_scratch_ptr[TM_MAX_DEVICES] = {};
std::mutex g_tm_device_lock[TM_MAX_DEVICES];

int tm_scratch_of_current_device(void **out, int *device_out) {
    int dev = -1;
    HIP_OK(hipGetDevice(&dev));
    if (dev < 0 || dev >= TM_MAX_DEVICES) {
        g_error = "rt_tonemap: device index out of range";
        return 10;
    }
    std::lock_guard<std::mutex> lk(g_tm_scratch_mutex);
    if (!g_tm_scratch_ptr[dev]) {
        void *ptr = nullptr;
        HIP_OK(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) {
            g_error = "rt_tonemap: no device address for the scratch array";
            return 10;
        }
        g_tm_scratch_ptr[dev] = ptr;
    }
    *out = g_tm_scratch_ptr[dev];
    *device_out = dev;
    return 0;
}

// The tone map of a frame in device memory into device memory, on the current device; allocates nothing.
int tm_run(const float *d_hdr, int32_t width, int32_t height, const rt_tonemap_desc *tm, uint8_t *d_out) {
    void *base = nullptr;
    int dev = 0;
    if (int rc = tm_scratch_of_current_device(&base, &dev)) return rc;
    double *partial = (double *)base;
    unsigned *state = (unsigned *)((unsigned char *)base + TM_SCRATCH_PARTIAL_BYTES), *hist = state + 4;
    const size_t n = (size_t)width * height;
    const int blocks = (int)std::min<size_t>(TM_BLOCKS, (n + TM_THREADS - 1) / TM_THREADS);
    TonemapParams p;
    p.tmo = tm->tmo;
    p.saturation = tm->saturation + 0.2f;
    p.inv_gamma = 1.0f / (tm->gamma > 1e-6f ? tm->gamma : 2.2f);
    p.use_state_white = tm->opt2 > 1e-6f ? 1 : 0;
    long long k = 0;
    if (p.use_state_white) {   // burn percentage -> the k-th smallest exposed luminance
        const float pct = (100.0f - tm->opt2) / 100.0f;
        k = (long long)std::floor(pct * (float)(n - 1));
        k = std::max<long long>(0, std::min<long long>((long long)n - 1, k));
    }
    std::lock_guard<std::mutex> lk(g_tm_device_lock[dev]);
    // the array keeps the previous call's contents: every block of the first launch overwrites its partial sum, and the
    // second sets the select state and clears the histogram before anything reads them
    hipLaunchKernelGGL(tm_luminance_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, d_hdr, n, (float *)nullptr, partial);
    hipLaunchKernelGGL(tm_prepare_kernel, dim3(1), dim3(64), 0, 0, partial, blocks, (double)n, tm->opt1, (unsigned)k, state, hist);
    if (p.use_state_white)
        for (int pass = 0; pass < 4; ++pass) {
            hipLaunchKernelGGL(tm_histogram_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, d_hdr, (const float *)nullptr, n, pass, state, hist);
            hipLaunchKernelGGL(tm_select_kernel, dim3(1), dim3(256), 0, 0, pass, state, hist);
        }
    hipLaunchKernelGGL(tm_apply_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, d_hdr, (const float *)nullptr, n, p, state, d_out);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(nullptr));
    return 0;
}

}  // namespace

extern "C" {

int rt_tonemap(const float *hdr, int32_t width, int32_t height, const rt_tonemap_desc *tm, uint8_t *rgb8_out) {
    if (!hdr || !tm || !rgb8_out || width < 1 || height < 1 || tm->tmo < RT_TMO_PHOTOGRAPHIC || tm->tmo > RT_TMO_ACES) {
        g_error = "rt_tonemap: bad argument";
        return 1;
    }
    const size_t n = (size_t)width * height;
    TmBuffers buf;   // device copies of buffers that live in host memory (the only allocations, and only then)
    const float *d_hdr = hdr;
    if (!tm_on_device(hdr)) {
        float *tmp = (float *)buf.get(n * 12);
        if (!tmp) {
            g_error = "rt_tonemap: out of device memory";
            return 10;
        }
        HIP_OK(hipMemcpy(tmp, hdr, n * 12, hipMemcpyHostToDevice));
        d_hdr = tmp;
    }
    const bool out_dev = tm_on_device(rgb8_out);
    uint8_t *d_out = out_dev ? rgb8_out : (uint8_t *)buf.get(n * 3);
    if (!d_out) {
        g_error = "rt_tonemap: out of device memory";
        return 10;
    }
    if (int rc = tm_run(d_hdr, width, height, tm, d_out)) return rc;
    if (!out_dev) HIP_OK(hipMemcpy(rgb8_out, d_out, n * 3, hipMemcpyDeviceToHost));
    return 0;
}

int rt_tonemap_noalloc(const float *hdr, int32_t width, int32_t height, const rt_tonemap_desc *tm, uint8_t *rgb8_out) {
    if (!hdr || !tm || !rgb8_out || width < 1 || height < 1 || tm->tmo < RT_TMO_PHOTOGRAPHIC || tm->tmo > RT_TMO_ACES) {
        g_error = "rt_tonemap_noalloc: bad argument";
        return 1;
    }
    if ((int64_t)width * height > RT_TONEMAP_NOALLOC_MAX_PIXELS) {
        g_error = "rt_tonemap_noalloc: frame larger than RT_TONEMAP_NOALLOC_MAX_PIXELS (2^28 pixels: ranks and histogram counts in the scratch buffer are 32-bit)";
        return 5;
    }
    int current = -1, dev_in = -1, dev_out = -1;
    HIP_OK(hipGetDevice(&current));
    if (!tm_on_device(hdr, &dev_in) || !tm_on_device(rgb8_out, &dev_out)) {
        g_error = "rt_tonemap_noalloc: hdr and rgb8_out must be device memory (staging host buffers would need an allocation)";
        return 1;
    }
    if (dev_in != current || dev_out != current) {   // the kernels run on the current device with that device's scratch array
        g_error = "rt_tonemap_noalloc: hdr and rgb8_out must be memory of the current device (hdr is on GPU " + std::to_string(dev_in) + ", rgb8_out on GPU " +
                  std::to_string(dev_out) + ", current is GPU " + std::to_string(current) + ")";
        return 1;
    }
    return tm_run(hdr, width, height, tm, rgb8_out);
}

int rt_quantize_ldr(const float *hdr, int64_t n_values, uint8_t *out) {
    if (!hdr || !out || n_values < 1) {
        g_error = "rt_quantize_ldr: bad argument";
        return 1;
    }
    const size_t n = (size_t)n_values;
    TmBuffers buf;
    const float *d_hdr = hdr;
    if (!tm_on_device(hdr)) {
        float *tmp = (float *)buf.get(n * 4);
        if (!tmp) {
            g_error = "rt_quantize_ldr: out of device memory";
            return 10;
        }
        HIP_OK(hipMemcpy(tmp, hdr, n * 4, hipMemcpyHostToDevice));
        d_hdr = tmp;
    }
    const bool out_dev = tm_on_device(out);
    uint8_t *d_out = out_dev ? out : (uint8_t *)buf.get(n);
    if (!d_out) {
        g_error = "rt_quantize_ldr: out of device memory";
        return 10;
    }
    const int blocks = (int)std::min<size_t>(TM_BLOCKS, (n + TM_THREADS - 1) / TM_THREADS);
    hipLaunchKernelGGL(tm_quantize_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, d_hdr, n, d_out);
    HIP_OK(hipGetLastError());
    if (!out_dev) HIP_OK(hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost));
    HIP_OK(hipDeviceSynchronize());
    return 0;
}

}  // extern "C"
