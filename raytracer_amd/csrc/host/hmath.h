// Host-side float vector / 4x4 matrix helpers used while ingesting a scene.  The summation order of
// dot products and matrix products follows the reference's SSE formulation
// ((a0*b0 + a1*b1) + (a2*b2 + a3*b3), reference src/algebra.cpp:31-35,143-244) so that matrices
// composed here carry the same rounding as the reference's.
#pragma once
#include <cmath>
#include <cstring>

namespace rth {

struct V3 {
    float x = 0, y = 0, z = 0;
    V3() {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
};

inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// reference normalize (src/algebra.cpp:50-65): zero vector when |v|^2 <= 1e-12
inline V3 normalize(V3 a) {
    float d = dot(a, a);
    if (!(d > 1e-12f)) return V3(0, 0, 0);
    float inv = 1.0f / sqrtf(d);
    return a * inv;
}
inline V3 vmin(V3 a, V3 b) { return V3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }
inline V3 vmax(V3 a, V3 b) { return V3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }

struct M4 {
    float m[16];
    M4() { identity(); }
    void identity() {
        memset(m, 0, sizeof m);
        m[0] = m[5] = m[10] = m[15] = 1.0f;
    }
    bool is_identity() const {
        static const M4 I;
        return memcmp(m, I.m, sizeof m) == 0;
    }
};

inline M4 translate(V3 t) {
    M4 r;
    r.m[3] = t.x;
    r.m[7] = t.y;
    r.m[11] = t.z;
    return r;
}
inline M4 scale(V3 s) {
    M4 r;
    r.m[0] = s.x;
    r.m[5] = s.y;
    r.m[10] = s.z;
    return r;
}
// Rodrigues rotation, angle in radians (reference src/algebra.cpp:99-117)
inline M4 rotate(V3 axis_in, float angle) {
    V3 a = normalize(axis_in);
    float x = a.x, y = a.y, z = a.z;
    float c = cosf(angle), s = sinf(angle), t = 1.0f - c;
    M4 r;
    r.m[0] = t * x * x + c;
    r.m[1] = t * x * y - s * z;
    r.m[2] = t * x * z + s * y;
    r.m[4] = t * x * y + s * z;
    r.m[5] = t * y * y + c;
    r.m[6] = t * y * z - s * x;
    r.m[8] = t * x * z - s * y;
    r.m[9] = t * y * z + s * x;
    r.m[10] = t * z * z + c;
    return r;
}
inline M4 mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i * 4 + j] = (a.m[i * 4 + 0] * b.m[0 * 4 + j] + a.m[i * 4 + 1] * b.m[1 * 4 + j]) +
                             (a.m[i * 4 + 2] * b.m[2 * 4 + j] + a.m[i * 4 + 3] * b.m[3 * 4 + j]);
    return r;
}
inline M4 transpose(const M4 &a) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i * 4 + j] = a.m[j * 4 + i];
    return r;
}
// The reciprocal of the determinant as the reference makes it (src/algebra.cpp:345-347): the SSE instruction's 12-bit
// estimate x0, then one Newton step 2 x0 - d x0^2, which lands at (1 - e^2) / d for an estimate that is off by e: never
// above the true quotient.  The estimate's table is implementation-defined; it is modelled here as 1 / d cut to 12
// significant bits (strictly below a quotient that has so few), which is what the reference's build gives on the CPU the
// fixtures were generated on for the determinants of untransformed, translated and power-of-two-scaled objects (1, 2,
// 0.5, ...: 1 / 1 -> 0.99975586 -> 0.99999994) and stays within 2 units in the last place of it for any other.
// That last bit matters: the inverse of the identity is (1 - 2^-24) I, the ray a triangle test sees is scaled by it
// (src/shapes.cpp:224-230), and whether a shadow ray towards a sample ON an emitter's triangle counts as blocked by
// that triangle (t < distance - epsilon, src/ray_trace.cpp:232-241) is decided at that level: with an exact inverse
// 16 % of such rays are, with the reference's 3 %.
inline float reference_rcp_det(float d) {
    const double q = 1.0 / (double)d;
    float f = (float)q;
    unsigned u;
    memcpy(&u, &f, 4);
    if (std::fabs((double)f) > std::fabs(q)) --u;   // towards zero
    unsigned t = u & ~0xfffu;
    float x0;
    memcpy(&x0, &t, 4);
    if (t == u && (double)x0 == q) {                // exact in 12 bits: the estimate stays below
        t -= 0x1000u;
        memcpy(&x0, &t, 4);
    }
    return (x0 + x0) - d * (x0 * x0);
}
// General 4x4 inverse by cofactors (evaluated in double, each rounded once) times the reference's reciprocal of the
// determinant (above).  The reference's SSE Cramer's rule (src/algebra.cpp:246-370) forms the cofactors in float;
// they are exact in both for identity, translation and axis scaling, and agree to float rounding otherwise.
inline M4 inverse(const M4 &a) {
    double s[16], inv[16];
    for (int i = 0; i < 16; ++i) s[i] = a.m[i];
    inv[0] = s[5] * s[10] * s[15] - s[5] * s[11] * s[14] - s[9] * s[6] * s[15] + s[9] * s[7] * s[14] + s[13] * s[6] * s[11] - s[13] * s[7] * s[10];
    inv[4] = -s[4] * s[10] * s[15] + s[4] * s[11] * s[14] + s[8] * s[6] * s[15] - s[8] * s[7] * s[14] - s[12] * s[6] * s[11] + s[12] * s[7] * s[10];
    inv[8] = s[4] * s[9] * s[15] - s[4] * s[11] * s[13] - s[8] * s[5] * s[15] + s[8] * s[7] * s[13] + s[12] * s[5] * s[11] - s[12] * s[7] * s[9];
    inv[12] = -s[4] * s[9] * s[14] + s[4] * s[10] * s[13] + s[8] * s[5] * s[14] - s[8] * s[6] * s[13] - s[12] * s[5] * s[10] + s[12] * s[6] * s[9];
    inv[1] = -s[1] * s[10] * s[15] + s[1] * s[11] * s[14] + s[9] * s[2] * s[15] - s[9] * s[3] * s[14] - s[13] * s[2] * s[11] + s[13] * s[3] * s[10];
    inv[5] = s[0] * s[10] * s[15] - s[0] * s[11] * s[14] - s[8] * s[2] * s[15] + s[8] * s[3] * s[14] + s[12] * s[2] * s[11] - s[12] * s[3] * s[10];
    inv[9] = -s[0] * s[9] * s[15] + s[0] * s[11] * s[13] + s[8] * s[1] * s[15] - s[8] * s[3] * s[13] - s[12] * s[1] * s[11] + s[12] * s[3] * s[9];
    inv[13] = s[0] * s[9] * s[14] - s[0] * s[10] * s[13] - s[8] * s[1] * s[14] + s[8] * s[2] * s[13] + s[12] * s[1] * s[10] - s[12] * s[2] * s[9];
    inv[2] = s[1] * s[6] * s[15] - s[1] * s[7] * s[14] - s[5] * s[2] * s[15] + s[5] * s[3] * s[14] + s[13] * s[2] * s[7] - s[13] * s[3] * s[6];
    inv[6] = -s[0] * s[6] * s[15] + s[0] * s[7] * s[14] + s[4] * s[2] * s[15] - s[4] * s[3] * s[14] - s[12] * s[2] * s[7] + s[12] * s[3] * s[6];
    inv[10] = s[0] * s[5] * s[15] - s[0] * s[7] * s[13] - s[4] * s[1] * s[15] + s[4] * s[3] * s[13] + s[12] * s[1] * s[7] - s[12] * s[3] * s[5];
    inv[14] = -s[0] * s[5] * s[14] + s[0] * s[6] * s[13] + s[4] * s[1] * s[14] - s[4] * s[2] * s[13] - s[12] * s[1] * s[6] + s[12] * s[2] * s[5];
    inv[3] = -s[1] * s[6] * s[11] + s[1] * s[7] * s[10] + s[5] * s[2] * s[11] - s[5] * s[3] * s[10] - s[9] * s[2] * s[7] + s[9] * s[3] * s[6];
    inv[7] = s[0] * s[6] * s[11] - s[0] * s[7] * s[10] - s[4] * s[2] * s[11] + s[4] * s[3] * s[10] + s[8] * s[2] * s[7] - s[8] * s[3] * s[6];
    inv[11] = -s[0] * s[5] * s[11] + s[0] * s[7] * s[9] + s[4] * s[1] * s[11] - s[4] * s[3] * s[9] - s[8] * s[1] * s[7] + s[8] * s[3] * s[5];
    inv[15] = s[0] * s[5] * s[10] - s[0] * s[6] * s[9] - s[4] * s[1] * s[10] + s[4] * s[2] * s[9] + s[8] * s[1] * s[6] - s[8] * s[2] * s[5];
    double det = s[0] * inv[0] + s[1] * inv[4] + s[2] * inv[8] + s[3] * inv[12];
    M4 r;
    if (det == 0.0) return r;
    const float id = reference_rcp_det((float)det);
    for (int i = 0; i < 16; ++i) r.m[i] = (float)inv[i] * id;
    return r;
}
// point (w=1) / direction (w=0) transform, reference src/algebra.cpp:217-244
inline V3 xform_point(const M4 &m, V3 v) {
    return V3((m.m[0] * v.x + m.m[1] * v.y) + (m.m[2] * v.z + m.m[3]), (m.m[4] * v.x + m.m[5] * v.y) + (m.m[6] * v.z + m.m[7]),
              (m.m[8] * v.x + m.m[9] * v.y) + (m.m[10] * v.z + m.m[11]));
}
inline V3 xform_dir(const M4 &m, V3 v) {
    return V3((m.m[0] * v.x + m.m[1] * v.y) + (m.m[2] * v.z + m.m[3] * 0.0f), (m.m[4] * v.x + m.m[5] * v.y) + (m.m[6] * v.z + m.m[7] * 0.0f),
              (m.m[8] * v.x + m.m[9] * v.y) + (m.m[10] * v.z + m.m[11] * 0.0f));
}

}  // namespace rth
