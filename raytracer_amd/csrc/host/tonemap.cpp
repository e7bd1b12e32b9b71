// Tone mapping of the HDR framebuffer to 8-bit RGB on the host (reference src/main.cpp:169-345):
// Reinhard photographic with key and burn-percent white point, Hable filmic, Narkowicz ACES;
// log-average luminance accumulated in double; saturation gets a +0.2 bias; gamma defaults to 2.2.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "gpu_render.h"
#include "tonemap.h"

namespace rth {

namespace {

float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

float aces_curve(float x) { return (x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f); }

float filmic_curve(float x) {
    const float A = 0.22f, B = 0.30f, C = 0.10f, D = 0.20f, E = 0.01f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - (E / F);
}

}  // namespace

void apply_tonemap(int tmo, float opt1, float opt2, float saturation_in, float gamma_in, int width, int height, const float *hdr,
                   uint8_t *out) {
    const size_t n = (size_t)width * height;
    std::vector<float> lum(n);
    for (size_t i = 0; i < n; ++i) lum[i] = 0.2126f * hdr[i * 3] + 0.7152f * hdr[i * 3 + 1] + 0.0722f * hdr[i * 3 + 2];

    float avg = 1.0f;
    if (n) {
        double sum_log = 0.0;
        for (float l : lum) sum_log += std::log(1e-4f + (l < 0.0f ? 0.0f : l));
        avg = (float)std::exp(sum_log / (double)n);
    }
    const float exposure = avg > 1e-6f ? opt1 / avg : 1.0f;

    float white = 1e9f;
    if (n && opt2 > 1e-6f) {   // burn percentage -> percentile of exposed luminance
        std::vector<float> e(lum);
        for (float &v : e) v = std::max(0.0f, v) * exposure;
        float pct = (100.0f - opt2) / 100.0f;
        int k = (int)std::floor(pct * (e.size() - 1));
        k = std::max(0, std::min((int)e.size() - 1, k));
        std::nth_element(e.begin(), e.begin() + k, e.end());
        white = std::max(e[k], 1e-6f);
    }
    const float white_sq = white * white;
    const float saturation = saturation_in + 0.2f;
    const float inv_gamma = 1.0f / (gamma_in > 1e-6f ? gamma_in : 2.2f);
    const float denom_aces = tmo == RT_TMO_ACES ? std::max(aces_curve(white), 1e-6f) : 1.0f;
    const float denom_filmic = tmo == RT_TMO_FILMIC ? std::max(filmic_curve(white), 1e-6f) : 1.0f;

    for (size_t i = 0; i < n; ++i) {
        const float y_in = lum[i];
        const float L = std::max(0.0f, y_in) * exposure;
        float y_out = 0.0f;
        if (tmo == RT_TMO_PHOTOGRAPHIC)
            y_out = (L * (1.0f + (L / white_sq))) / (1.0f + L);
        else if (tmo == RT_TMO_ACES)
            y_out = aces_curve(L) / denom_aces;
        else if (tmo == RT_TMO_FILMIC)
            y_out = filmic_curve(L) / denom_filmic;
        y_out = clamp01(y_out);
        float c[3] = {0.f, 0.f, 0.f};
        if (!(y_in < 1e-6f)) {
            for (int k = 0; k < 3; ++k) {
                float ratio = hdr[i * 3 + k] / y_in;
                if (std::abs(saturation - 1.0f) > 1e-4f) ratio = std::pow(std::max(0.0f, ratio), saturation);
                c[k] = y_out * ratio;
            }
        }
        for (int k = 0; k < 3; ++k) out[i * 3 + k] = (uint8_t)std::lround(255.0f * std::pow(clamp01(c[k]), inv_gamma));
    }
}

}  // namespace rth
