// TEST INFRASTRUCTURE — part of the CPU oracle; never linked into the product.
//
// Deterministic single-precision elementary functions built only from IEEE-754 add, subtract,
// multiply, divide, sqrt and integer bit manipulation (Cephes-style argument reduction and
// minimax polynomials).  The reference calls libm (std::cos, powf, std::acos, ...) and the
// vendored sse_mathfun exp_ps (reference src/algebra.cpp:85-88, third_party/sse_mathfun); those
// differ between libm builds and from a GPU math library in the last bits, which would make
// secondary rays diverge between two implementations that are otherwise identical.  Stating the
// functions at this level lets the oracle and the device code evaluate the same operation
// sequence, so that every ray — not only primary rays — can be compared exactly.  Accuracy is
// 1-3 ulp, i.e. the same order as the libm / sse_mathfun differences of the reference itself.
// Compile with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace om {

inline uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
inline float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline float absf(float x) { return u2f(f2u(x) & 0x7fffffffu); }

// sin and cos share the Cody-Waite reduction by pi/4 octants (valid for |x| < 8192).
inline void reduce_octant(float ax, int &j, float &r) {
    j = (int)(ax * 1.27323954473516f);   // 4/pi
    if (j & 1) j += 1;
    float y = (float)j;
    r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}
inline float sin_poly(float x) {
    float z = x * x;
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
}
inline float cos_poly(float x) {
    float z = x * x;
    return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}
inline float sinf_(float x) {
    bool neg = x < 0.0f;
    float ax = absf(x);
    int j;
    float r;
    reduce_octant(ax, j, r);
    j &= 7;
    if (j > 3) {
        neg = !neg;
        j -= 4;
    }
    float y = (j == 1 || j == 2) ? cos_poly(r) : sin_poly(r);
    return neg ? -y : y;
}
inline float cosf_(float x) {
    bool neg = false;
    float ax = absf(x);
    int j;
    float r;
    reduce_octant(ax, j, r);
    j &= 7;
    if (j > 3) {
        j -= 4;
        neg = !neg;
    }
    if (j > 1) neg = !neg;
    float y = (j == 1 || j == 2) ? sin_poly(r) : cos_poly(r);
    return neg ? -y : y;
}

inline float asinf_(float x) {
    bool neg = x < 0.0f;
    float a = absf(x);
    if (a > 1.0f) return 0.0f;
    if (a < 1.0e-4f) return x;
    bool flag = a > 0.5f;
    float z, xx;
    if (flag) {
        z = 0.5f * (1.0f - a);
        xx = sqrtf(z);
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
inline float acosf_(float x) {
    if (x < -1.0f || x > 1.0f) return 0.0f;
    if (x > 0.5f) return 2.0f * asinf_(sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asinf_(sqrtf(0.5f * (1.0f + x)));
    return 1.5707963267948966192f - asinf_(x);
}
inline float atanf_(float x) {
    bool neg = x < 0.0f;
    float a = absf(x);
    float y;
    if (a > 2.414213562373095f) {
        y = 1.5707963267948966192f;
        a = -(1.0f / a);
    } else if (a > 0.4142135623730950f) {
        y = 0.7853981633974483096f;
        a = (a - 1.0f) / (a + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = a * a;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * a + a);
    return neg ? -y : y;
}
inline float atan2f_(float y, float x) {
    const float pi = 3.14159265358979323846f, half_pi = 1.5707963267948966192f;
    if (x == 0.0f) {
        if (y > 0.0f) return half_pi;
        if (y < 0.0f) return -half_pi;
        return 0.0f;
    }
    float z = atanf_(y / x);
    if (x < 0.0f) return (y < 0.0f) ? z - pi : z + pi;
    return z;
}

// z * 2^n for normal results, 0 on underflow, +inf on overflow
inline float ldexpf_(float z, int n) {
    if (n > 127) {
        z = z * 1.7014118346046923e38f;   // 2^127
        n -= 127;
        if (n > 127) n = 127;
    } else if (n < -126) {
        z = z * 1.1754943508222875e-38f;   // 2^-126
        n += 126;
        if (n < -126) n = -126;
    }
    return z * u2f((uint32_t)(n + 127) << 23);
}
inline float expf_(float x) {
    if (x > 88.72283905206835f) return u2f(0x7f800000u);
    if (x < -103.278929903431851103f) return 0.0f;
    float z = floorf(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return ldexpf_(z, n);
}
// natural logarithm of a positive normal float
inline float logf_(float x) {
    uint32_t u = f2u(x);
    int e = (int)((u >> 23) & 0xff) - 126;
    x = u2f((u & 0x007fffffu) | 0x3f000000u);   // mantissa in [0.5, 1)
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
// x^y for x >= 0 (every call site on the path clamps the base to >= 0)
inline float powf_(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (!(x > 1.1754943508222875e-38f)) return 0.0f;   // zero, denormal, negative or NaN base
    return expf_(y * logf_(x));
}

}  // namespace om
