// Tone mapping and 8-bit quantisation of the HDR framebuffer on the device: apply_tonemap and the plain LDR
// conversion of the reference (src/main.cpp:169-345, :579-583), so that a frame that is only wanted as PNG never
// leaves HBM as floats.  Included by render.hip (shares its error string and HIP_OK).
//
// Steps: luminance + log-average (per-thread double sums, fixed-order tree per block, fixed-order sum of the block
// results: the same value every run), exposure, the burn-percent white point as an exact order statistic (radix
// select over the float bit patterns, four 8-bit passes), then the per-pixel curve, saturation and gamma.
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
__global__ void __launch_bounds__(TM_THREADS) tm_histogram_kernel(const float *__restrict__ hdr, const float *__restrict__ lum, size_t n, float exposure, int pass,
                                                                   const unsigned *__restrict__ state, unsigned *__restrict__ hist) {
    __shared__ unsigned local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
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

__global__ void tm_select_kernel(int pass, unsigned *__restrict__ state, unsigned *__restrict__ hist) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int shift = 24 - 8 * pass;
    unsigned k = state[1], bin = 255;
    for (unsigned b = 0; b < 256; ++b) {
        const unsigned c = hist[b];
        if (k < c) {
            bin = b;
            break;
        }
        k -= c;
    }
    for (unsigned b = 0; b < 256; ++b) hist[b] = 0;
    state[0] |= bin << shift;
    state[1] = k;
    if (pass == 3) state[2] = state[0];
}

struct TonemapParams {
    int tmo;
    float exposure, saturation, inv_gamma;
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
    float denom = 1.0f;
    if (p.tmo == RT_TMO_ACES) denom = fmaxf(tm_aces(white), 1e-6f);
    if (p.tmo == RT_TMO_FILMIC) denom = fmaxf(tm_filmic(white), 1e-6f);
    for (size_t i = (size_t)blockIdx.x * TM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * TM_THREADS) {
        const float y_in = lum ? lum[i] : tm_luminance(hdr + 3 * i);
        const float L = (y_in < 0.0f ? 0.0f : y_in) * p.exposure;
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

// ---- scratch of rt_tonemap_noalloc ------------------------------------------------------------------------------
// Everything rt_tonemap allocates per call except the luminance plane (which the no-allocation variant does without):
// TM_BLOCKS double partial sums, then the radix select's state (4 words) and histogram (256 words).  One copy per GPU
// for the life of the process; calls that overlap on one GPU would race on it.
constexpr size_t TM_SCRATCH_PARTIAL_BYTES = TM_BLOCKS * sizeof(double);
constexpr size_t TM_SCRATCH_BYTES = TM_SCRATCH_PARTIAL_BYTES + (4 + 256) * sizeof(unsigned);
static __device__ __align__(256) unsigned char scratch[(TM_SCRATCH_BYTES + 255) / 256 * 256];

// The device address of `scratch` on the current device.  In host code the array's name is only a handle for
// hipGetSymbolAddress; every GPU holds its own copy at its own address, looked up once per device.
constexpr int TM_MAX_DEVICES = 64;
std::mutex g_tm_scratch_mutex;
void *g_tm_scratch_ptr[TM_MAX_DEVICES] = {};

int tm_scratch_of_current_device(void **out) {
    int dev = -1;
    HIP_OK(hipGetDevice(&dev));
    if (dev < 0 || dev >= TM_MAX_DEVICES) {
        g_error = "rt_tonemap_noalloc: device index out of range";
        return 10;
    }
    std::lock_guard<std::mutex> lk(g_tm_scratch_mutex);
    if (!g_tm_scratch_ptr[dev]) {
        void *ptr = nullptr;
        HIP_OK(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) {
            g_error = "rt_tonemap_noalloc: no device address for the scratch array";
            return 10;
        }
        g_tm_scratch_ptr[dev] = ptr;
    }
    *out = g_tm_scratch_ptr[dev];
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
    TmBuffers buf;
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
    float *lum = (float *)buf.get(n * 4);
    double *partial = (double *)buf.get(TM_BLOCKS * sizeof(double));
    unsigned *state = (unsigned *)buf.get((4 + 256) * sizeof(unsigned)), *hist = state + 4;
    if (!d_out || !lum || !partial || !state) {
        g_error = "rt_tonemap: out of device memory";
        return 10;
    }
    const int blocks = (int)std::min<size_t>(TM_BLOCKS, (n + TM_THREADS - 1) / TM_THREADS);
    hipLaunchKernelGGL(tm_luminance_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, d_hdr, n, lum, partial);
    HIP_OK(hipGetLastError());
    double host_partial[TM_BLOCKS];
    HIP_OK(hipMemcpy(host_partial, partial, blocks * sizeof(double), hipMemcpyDeviceToHost));
    double sum_log = 0.0;
    for (int b = 0; b < blocks; ++b) sum_log += host_partial[b];
    const float avg = (float)std::exp(sum_log / (double)n);
    TonemapParams p;
    p.tmo = tm->tmo;
    p.exposure = avg > 1e-6f ? tm->opt1 / avg : 1.0f;
    p.saturation = tm->saturation + 0.2f;
    p.inv_gamma = 1.0f / (tm->gamma > 1e-6f ? tm->gamma : 2.2f);
    p.use_state_white = tm->opt2 > 1e-6f ? 1 : 0;
    if (p.use_state_white) {   // burn percentage -> the k-th smallest exposed luminance
        const float pct = (100.0f - tm->opt2) / 100.0f;
        long long k = (long long)std::floor(pct * (float)(n - 1));
        k = std::max<long long>(0, std::min<long long>((long long)n - 1, k));
        const unsigned init[4] = {0u, (unsigned)k, 0u, 0u};
        HIP_OK(hipMemcpy(state, init, sizeof init, hipMemcpyHostToDevice));
        HIP_OK(hipMemset(hist, 0, 256 * sizeof(unsigned)));
        for (int pass = 0; pass < 4; ++pass) {
            hipLaunchKernelGGL(tm_histogram_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, d_hdr, lum, n, p.exposure, pass, state, hist);
            hipLaunchKernelGGL(tm_select_kernel, dim3(1), dim3(1), 0, 0, pass, state, hist);
        }
    }
    hipLaunchKernelGGL(tm_apply_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, d_hdr, lum, n, p, state, d_out);
    HIP_OK(hipGetLastError());
    if (!out_dev) HIP_OK(hipMemcpy(rgb8_out, d_out, n * 3, hipMemcpyDeviceToHost));
    HIP_OK(hipDeviceSynchronize());
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
    void *base = nullptr;
    if (int rc = tm_scratch_of_current_device(&base)) return rc;
    double *partial = (double *)base;
    unsigned *state = (unsigned *)((unsigned char *)base + TM_SCRATCH_PARTIAL_BYTES), *hist = state + 4;
    const size_t n = (size_t)width * height;
    const int blocks = (int)std::min<size_t>(TM_BLOCKS, (n + TM_THREADS - 1) / TM_THREADS);
    // the array keeps the previous call's contents: every block of this launch overwrites its partial sum, and the
    // select state and histogram are set below before they are read
    hipLaunchKernelGGL(tm_luminance_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, hdr, n, (float *)nullptr, partial);
    HIP_OK(hipGetLastError());
    double host_partial[TM_BLOCKS];
    HIP_OK(hipMemcpy(host_partial, partial, blocks * sizeof(double), hipMemcpyDeviceToHost));
    double sum_log = 0.0;
    for (int b = 0; b < blocks; ++b) sum_log += host_partial[b];
    const float avg = (float)std::exp(sum_log / (double)n);
    TonemapParams p;
    p.tmo = tm->tmo;
    p.exposure = avg > 1e-6f ? tm->opt1 / avg : 1.0f;
    p.saturation = tm->saturation + 0.2f;
    p.inv_gamma = 1.0f / (tm->gamma > 1e-6f ? tm->gamma : 2.2f);
    p.use_state_white = tm->opt2 > 1e-6f ? 1 : 0;
    if (p.use_state_white) {
        const float pct = (100.0f - tm->opt2) / 100.0f;
        long long k = (long long)std::floor(pct * (float)(n - 1));
        k = std::max<long long>(0, std::min<long long>((long long)n - 1, k));
        const unsigned init[4] = {0u, (unsigned)k, 0u, 0u};
        HIP_OK(hipMemcpy(state, init, sizeof init, hipMemcpyHostToDevice));
        HIP_OK(hipMemsetAsync(hist, 0, 256 * sizeof(unsigned), 0));
        for (int pass = 0; pass < 4; ++pass) {
            hipLaunchKernelGGL(tm_histogram_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, hdr, (const float *)nullptr, n, p.exposure, pass, state, hist);
            hipLaunchKernelGGL(tm_select_kernel, dim3(1), dim3(1), 0, 0, pass, state, hist);
        }
    }
    hipLaunchKernelGGL(tm_apply_kernel, dim3(blocks), dim3(TM_THREADS), 0, 0, hdr, (const float *)nullptr, n, p, state, rgb8_out);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    return 0;
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
