// Device arithmetic of the rendering core: vec3 helpers in a fixed evaluation order and
// deterministic single-precision elementary functions (argument reduction + minimax polynomials
// built only from IEEE add/sub/mul/div/sqrt and integer bit operations).
//
// The evaluation order of every expression here is part of the renderer's definition: dot products
// sum as (x*x' + y*y') + z*z' (the reference's _mm_dp_ps order, reference src/algebra.cpp:31-35),
// matrix-vector products as (m0*x + m1*y) + (m2*z + m3*w) (src/algebra.cpp:217-244).  The file must
// be compiled with -ffp-contract=off so that no multiply-add is fused behind the source's back;
// with that, a strict-IEEE CPU evaluation of the same expressions gives bit-identical results,
// which is how primary-hit identities are checked.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct V {
    float x, y, z;
};
__device__ __forceinline__ V mk(float a, float b, float c) { return V{a, b, c}; }
__device__ __forceinline__ V mk(const float *p) { return V{p[0], p[1], p[2]}; }
__device__ __forceinline__ V add(V a, V b) { return V{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V sub(V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V mul(V a, V b) { return V{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V scl(V a, float s) { return V{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V a, V b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V cross(V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float rt_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float rt_div(float a, float b) { return a / b; }
// reference normalize (src/algebra.cpp:50-65) with an exact reciprocal square root; zero vector when |a|^2 <= 1e-12
__device__ __forceinline__ V normalize(V a) {
    float d = dot(a, a);
    if (!(d > 1e-12f)) return V{0.f, 0.f, 0.f};
    float inv = rt_div(1.0f, rt_sqrt(d));
    return scl(a, inv);
}
__device__ __forceinline__ float max_std(float a, float b) { return a < b ? b : a; }   // std::max
__device__ __forceinline__ float min_std(float a, float b) { return b < a ? b : a; }   // std::min
__device__ __forceinline__ float absf(float x) { return __uint_as_float(__float_as_uint(x) & 0x7fffffffu); }
__device__ __forceinline__ V xpoint(const float *m, V v) {
    return V{(m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3]), (m[4] * v.x + m[5] * v.y) + (m[6] * v.z + m[7]),
             (m[8] * v.x + m[9] * v.y) + (m[10] * v.z + m[11])};
}
__device__ __forceinline__ V xdir(const float *m, V v) {
    return V{(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z, (m[8] * v.x + m[9] * v.y) + m[10] * v.z};
}

// ---- counter-based random numbers --------------------------------------------------------------
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t sample_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return hash32(hash32(hash32(seed + 0x9e3779b9u) ^ pixel) ^ (sample * 0x85ebca6bu + 1u));
}
__device__ __forceinline__ uint32_t child_key(uint32_t key, uint32_t branch) { return hash32(key + 0x9e3779b9u * (branch + 1u)); }
__device__ __forceinline__ float rnd(uint32_t key, uint32_t dim) {
    return (float)(hash32(key ^ hash32(dim + 0x68bc21ebu)) >> 8) * (1.0f / 16777216.0f);
}

// ---- elementary functions ------------------------------------------------------------------------
__device__ __forceinline__ void reduce_octant(float ax, int &j, float &r) {
    j = (int)(ax * 1.27323954473516f);
    if (j & 1) j += 1;
    float y = (float)j;
    r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}
__device__ __forceinline__ float sin_poly(float x) {
    float z = x * x;
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
}
__device__ __forceinline__ float cos_poly(float x) {
    float z = x * x;
    return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}
__device__ inline float rt_sin(float x) {
    bool neg = x < 0.0f;
    int j;
    float r;
    reduce_octant(absf(x), j, r);
    j &= 7;
    if (j > 3) {
        neg = !neg;
        j -= 4;
    }
    float y = (j == 1 || j == 2) ? cos_poly(r) : sin_poly(r);
    return neg ? -y : y;
}
__device__ inline float rt_cos(float x) {
    bool neg = false;
    int j;
    float r;
    reduce_octant(absf(x), j, r);
    j &= 7;
    if (j > 3) {
        j -= 4;
        neg = !neg;
    }
    if (j > 1) neg = !neg;
    float y = (j == 1 || j == 2) ? sin_poly(r) : cos_poly(r);
    return neg ? -y : y;
}
// Sine and cosine of two angles at once.  Element for element the operations, and so the results, are those of rt_sin
// and rt_cos: one argument reduction serves both functions (they reduce the same |x|), both polynomials are evaluated
// for both elements with two-wide packed multiplies and adds, and each result picks its polynomial and sign.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 sin_poly2(f2 x) {
    const f2 z = x * x;
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
}
__device__ __forceinline__ f2 cos_poly2(f2 x) {
    const f2 z = x * x;
    return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}
__device__ __forceinline__ void rt_sincos2(f2 x, f2 &sn, f2 &cs) {
    const f2 ax = f2{absf(x.x), absf(x.y)};
    const f2 q = ax * 1.27323954473516f;
    int j0 = (int)q.x, j1 = (int)q.y;
    j0 += j0 & 1;
    j1 += j1 & 1;
    const f2 y = f2{(float)j0, (float)j1};
    const f2 r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const f2 sp = sin_poly2(r), cp = cos_poly2(r);
    j0 &= 7;
    j1 &= 7;
    // rt_sin: negative argument or octants 4..7 flip the sign; octants 1, 2 (mod 4) take the cosine polynomial
    // rt_cos: octants 4..7 flip, then octants 2, 3 (mod 4) flip; octants 1, 2 (mod 4) take the sine polynomial
    const bool hi0 = j0 > 3, hi1 = j1 > 3;
    const int k0 = hi0 ? j0 - 4 : j0, k1 = hi1 ? j1 - 4 : j1;
    const bool mid0 = k0 == 1 || k0 == 2, mid1 = k1 == 1 || k1 == 2;
    const float s0 = mid0 ? cp.x : sp.x, s1 = mid1 ? cp.y : sp.y;
    const float c0 = mid0 ? sp.x : cp.x, c1 = mid1 ? sp.y : cp.y;
    const bool sneg0 = (x.x < 0.0f) != hi0, sneg1 = (x.y < 0.0f) != hi1;
    const bool cneg0 = hi0 != (k0 > 1), cneg1 = hi1 != (k1 > 1);
    sn = f2{sneg0 ? -s0 : s0, sneg1 ? -s1 : s1};
    cs = f2{cneg0 ? -c0 : c0, cneg1 ? -c1 : c1};
}
__device__ inline float rt_asin(float x) {
    bool neg = x < 0.0f;
    float a = absf(x);
    if (a > 1.0f) return 0.0f;
    if (a < 1.0e-4f) return x;
    bool flag = a > 0.5f;
    float z, xx;
    if (flag) {
        z = 0.5f * (1.0f - a);
        xx = rt_sqrt(z);
    } else {
        xx = a;
        z = xx * xx;
    }
    z = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * xx + xx;
    if (flag) {
        z = z + z;
        z = 1.5707963267948966192f - z;
    }
    return neg ? -z : z;
}
__device__ inline float rt_acos(float x) {
    if (x < -1.0f || x > 1.0f) return 0.0f;
    if (x > 0.5f) return 2.0f * rt_asin(rt_sqrt(0.5f * (1.0f - x)));
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * rt_asin(rt_sqrt(0.5f * (1.0f + x)));
    return 1.5707963267948966192f - rt_asin(x);
}
__device__ inline float rt_atan(float x) {
    bool neg = x < 0.0f;
    float a = absf(x);
    float y;
    if (a > 2.414213562373095f) {
        y = 1.5707963267948966192f;
        a = -rt_div(1.0f, a);
    } else if (a > 0.4142135623730950f) {
        y = 0.7853981633974483096f;
        a = rt_div(a - 1.0f, a + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = a * a;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * a + a);
    return neg ? -y : y;
}
__device__ inline float rt_atan2(float y, float x) {
    const float pi = 3.14159265358979323846f, half_pi = 1.5707963267948966192f;
    if (x == 0.0f) {
        if (y > 0.0f) return half_pi;
        if (y < 0.0f) return -half_pi;
        return 0.0f;
    }
    float z = rt_atan(rt_div(y, x));
    if (x < 0.0f) return (y < 0.0f) ? z - pi : z + pi;
    return z;
}
__device__ __forceinline__ float rt_ldexp(float z, int n) {
    if (n > 127) {
        z = z * 1.7014118346046923e38f;
        n -= 127;
        if (n > 127) n = 127;
    } else if (n < -126) {
        z = z * 1.1754943508222875e-38f;
        n += 126;
        if (n < -126) n = -126;
    }
    return z * __uint_as_float((uint32_t)(n + 127) << 23);
}
__device__ inline float rt_exp(float x) {
    if (x > 88.72283905206835f) return __uint_as_float(0x7f800000u);
    if (x < -103.278929903431851103f) return 0.0f;
    float z = floorf(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return rt_ldexp(z, n);
}
__device__ inline float rt_log(float x) {
    uint32_t u = __float_as_uint(x);
    int e = (int)((u >> 23) & 0xff) - 126;
    x = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (x < 0.707106781186547524f) {
        e -= 1;
        x = x + x - 1.0f;
    } else {
        x = x - 1.0f;
    }
    float z = x * x;
    float y = ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x - 1.2420140846e-1f) * x + 1.4249322787e-1f) * x - 1.6668057665e-1f) * x + 2.0000714765e-1f) * x - 2.4999993993e-1f) * x + 3.3333331174e-1f) * x * z;
    float fe = (float)e;
    if (e) y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = x + y;
    if (e) z = z + 0.693359375f * fe;
    return z;
}
// x^y for x >= 0
__device__ inline float rt_pow(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (!(x > 1.1754943508222875e-38f)) return 0.0f;
    return rt_exp(y * rt_log(x));
}

}  // namespace rt
