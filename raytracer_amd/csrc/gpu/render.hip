// MI355X (gfx950, wave64) rendering core: C ABI of include/gpu_render.h and the HIP kernels behind it.
//
// What is replaced: the reference's per-camera std::thread scheduler and everything it calls at
// render time (reference src/main.cpp:437-555 -> camera::get_samples src/camera.cpp:117-224,
// ray_tracer::trace src/ray_trace.cpp:1141-1172, bvh::intersect src/bvh.cpp:383-480, the shape
// tests src/shapes.cpp:212,407,506, calculate_color src/ray_trace.cpp:154-406, Light::get_sample
// src/lights.cpp, texture::sample src/texture.cpp:168-197).
//
// Execution shape: one lane per pixel, 8x8 pixel sub-tiles per wave, 16x16 per workgroup, workgroups
// enumerated over 32x32 image tiles (the reference's square_dynamic tile, src/main.cpp:441) so that a
// frame can be sharded by tile over ranks.  A lane walks its pixel's samples in order and, per
// sample, its path tree depth-first through a small explicit stack of pending rays carrying their
// throughput (the reference's recursion, flattened).  Results of a pixel therefore do not depend on
// launch geometry, sharding or scheduling, and sample sums are taken in the reference's order.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "gpu_render.h"
#include "rt_math.h"

using namespace rt;

namespace {

thread_local std::string g_error;

#define HIP_OK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            g_error = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
            return 10;                                                                                \
        }                                                                                             \
    } while (0)

constexpr int TILE = 32;              // image tile edge used for sharding
constexpr int BLOCK_EDGE = 16;        // pixels per workgroup edge
constexpr int PATH_STACK = 24;        // pending rays per lane
constexpr int TRAVERSAL_STACK = 64;   // src/bvh.cpp:423

enum {
    DIM_REFLECT_XI1 = 0, DIM_REFLECT_XI2 = 1, DIM_REFRACT_XI1 = 2, DIM_REFRACT_XI2 = 3, DIM_RR = 4, DIM_HEMI_1 = 5, DIM_HEMI_2 = 6,
    DIM_SPLIT = 8, DIM_LIGHT = 4096,
    DIM_CAM_R1 = 0, DIM_CAM_R2 = 1, DIM_CAM_R3 = 2, DIM_CAM_R4 = 3, DIM_CAM_TIME = 4,
    BRANCH_ROOT = 0, BRANCH_REFLECT = 1, BRANCH_REFRACT = 2, BRANCH_BOUNCE = 3
};

// ---- device-resident scene -------------------------------------------------------------------
// nodes:  2 x float4 per BVH node  {bmin.xyz, bits(offset)} {bmax.xyz, bits(n_prims | axis << 16)}
// prims:  3 x float4 per leaf entry, in leaf order:
//         triangle {c1s.xyz, e1.x} {e1.yz, e2.xy} {e2.z, bits(instance), bits(shape), bits(ref id)}
//         sphere   {...unused...}                 {0, bits(-1 - sphere), bits(shape), bits(123456)}
// inst:   4 x float4 per placement {inv row0} {inv row1} {inv row2} {motionblur.xyz, bits(flags)}
//         flags bit0 = identity transform, bit1 = has motion blur
struct DevScene {
    const float4 *nodes;
    const float4 *prims;
    const float4 *inst;
    const rt_instance *instances;
    const rt_triangle *triangles;
    const float *flat_normals;
    const rt_sphere *spheres;
    const rt_plane *planes;
    const rt_material *materials;
    const rt_texture *textures;
    const int32_t *texture_refs;
    const rt_image *images;
    const uint8_t *texels_u8;
    const float *texels_f32;
    const rt_light *lights;
    int n_triangles, n_spheres, n_planes, n_lights, n_nodes;
    float ieps, seps, max_depth;
    float ambient[3], bg[3];
    int bg_tex, is_probe;
};

struct Counters {   // per-lane, reduced per wave at kernel end
    uint32_t primary, shadow, secondary, shaded;
    uint32_t nodes, tri, sph, pln, texels;
};

struct Hit {
    float t;
    int prim;   // >= 0 leaf entry; <= -2 plane (-2 - index); -1 none
};

// ---- traversal ---------------------------------------------------------------------------------
// bvh::intersect (src/bvh.cpp:383-480) with intersect_aabb_fast (:348-381): planes first, then an
// ordered stack traversal (near child first by the sign of the ray direction on the split axis).
// A hit replaces the current one only if t > eps and t < t_hit, so of equal-distance hits the one
// found first in this order is kept.
template <bool ANY, bool COUNT>
__device__ Hit traverse(const DevScene &S, V o, V d, float time, bool culling, float eps, float stop_t, Counters &c) {
    Hit h;
    h.t = stop_t;
    h.prim = -1;
    for (int pi = 0; pi < S.n_planes; ++pi) {   // plane::intersect, src/shapes.cpp:506-527
        if (COUNT) ++c.pln;
        const rt_plane &p = S.planes[pi];
        V n = mk(p.normal);
        float denom = dot(n, d);
        if (absf(denom) < eps) continue;
        float num = dot(sub(mk(p.point), o), n);
        float tc = rt_div(num, denom);
        if (tc < eps) continue;
        if (tc > eps && tc < h.t) {
            h.t = tc;
            h.prim = -2 - pi;
            if (ANY) return h;
        }
    }
    if (S.n_nodes == 0) return h;
    const float ix = rt_div(1.0f, d.x), iy = rt_div(1.0f, d.y), iz = rt_div(1.0f, d.z);
    const bool nx = ix < 0, ny = iy < 0, nz = iz < 0;
    int stack[TRAVERSAL_STACK];
    int top = 0;
    stack[0] = 0;
    int cur_inst = -1;   // placement whose object-space ray is cached in oo/dd
    V oo = o, dd = d;
    while (top >= 0) {
        const int ni = stack[top--];
        const float4 n0 = S.nodes[2 * ni], n1 = S.nodes[2 * ni + 1];
        if (COUNT) ++c.nodes;
        float tmin = ((nx ? n1.x : n0.x) - o.x) * ix;
        float tmax = ((nx ? n0.x : n1.x) - o.x) * ix;
        const float tymin = ((ny ? n1.y : n0.y) - o.y) * iy;
        const float tymax = ((ny ? n0.y : n1.y) - o.y) * iy;
        if ((tmin > tymax) || (tymin > tmax)) continue;
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        const float tzmin = ((nz ? n1.z : n0.z) - o.z) * iz;
        const float tzmax = ((nz ? n0.z : n1.z) - o.z) * iz;
        if ((tmin > tzmax) || (tzmin > tmax)) continue;
        if (tzmin > tmin) tmin = tzmin;
        if (tzmax < tmax) tmax = tzmax;
        if (!((tmin < h.t) && (tmax > 0.0f))) continue;
        const int offset = __float_as_int(n0.w);
        const int meta = __float_as_int(n1.w);
        const int nprims = meta & 0xffff;
        if (nprims > 0) {
            for (int i = 0; i < nprims; ++i) {
                const int k = offset + i;
                const float4 p2 = S.prims[3 * k + 2];
                const int inst = __float_as_int(p2.y);
                float tc;
                if (inst >= 0) {   // triangle::intersect, src/shapes.cpp:212-279
                    if (COUNT) ++c.tri;
                    const float4 p0 = S.prims[3 * k], p1 = S.prims[3 * k + 1];
                    if (inst != cur_inst) {
                        cur_inst = inst;
                        const float4 r3 = S.inst[4 * inst + 3];
                        const int flags = __float_as_int(r3.w);
                        V moved = o;
                        if (flags & 2) moved = sub(o, scl(mk(r3.x, r3.y, r3.z), time));
                        if (flags & 1) {
                            oo = moved;
                            dd = d;
                        } else {
                            const float4 r0 = S.inst[4 * inst], r1 = S.inst[4 * inst + 1], r2 = S.inst[4 * inst + 2];
                            oo = mk((r0.x * moved.x + r0.y * moved.y) + (r0.z * moved.z + r0.w),
                                    (r1.x * moved.x + r1.y * moved.y) + (r1.z * moved.z + r1.w),
                                    (r2.x * moved.x + r2.y * moved.y) + (r2.z * moved.z + r2.w));
                            dd = mk((r0.x * d.x + r0.y * d.y) + r0.z * d.z, (r1.x * d.x + r1.y * d.y) + r1.z * d.z,
                                    (r2.x * d.x + r2.y * d.y) + r2.z * d.z);
                        }
                    }
                    const V c1s = mk(p0.x, p0.y, p0.z), e1 = mk(p0.w, p1.x, p1.y), e2 = mk(p1.z, p1.w, p2.x);
                    const V os = scl(oo, 1000.0f);
                    const V hh = cross(dd, e2);
                    const float det = dot(e1, hh);
                    if (culling) {
                        if (det < eps) continue;
                    } else {
                        if (absf(det) < eps) continue;
                    }
                    const float f = rt_div(1.0f, det);
                    const V s = sub(os, c1s);
                    float u = dot(s, hh);
                    u *= f;
                    if (u < (0.0f - eps) || u > (1.0f + eps)) continue;
                    const V q = cross(s, e1);
                    float v = dot(dd, q);
                    v *= f;
                    if (v < (0.0f - eps) || (u + v) > (1.0f + eps)) continue;
                    tc = dot(e2, q);
                    tc *= f;
                    tc *= (1.0f / 1000.0f);
                    if (!(tc > eps)) continue;
                } else {   // sphere::intersect, src/shapes.cpp:407-447
                    if (COUNT) ++c.sph;
                    const rt_sphere &sp = S.spheres[-1 - inst];
                    const V so = xpoint(sp.inv_model, o);
                    const V sd = xdir(sp.inv_model, d);
                    const V oc = sub(so, mk(sp.center));
                    const float a = dot(sd, sd);
                    float b = dot(oc, sd);
                    b *= 2.0f;
                    float cc = dot(oc, oc);
                    cc -= (sp.radius * sp.radius);
                    const float disc = b * b - 4 * a * cc;
                    if (disc < 0.0f) continue;
                    const float sq = rt_sqrt(disc);
                    const float inv2a = rt_div(1.0f, 2.0f * a);
                    const float t1 = (-b - sq) * inv2a, t2 = (-b + sq) * inv2a;
                    if (t1 > eps)
                        tc = t1;
                    else if (t2 > eps)
                        tc = t2;
                    else
                        continue;
                }
                if (tc > eps && tc < h.t) {
                    h.t = tc;
                    h.prim = k;
                    if (ANY) return h;
                }
            }
        } else {
            const int axis = meta >> 16;
            const bool neg = axis == 0 ? nx : (axis == 1 ? ny : nz);
            const int near_i = neg ? offset : ni + 1;
            const int far_i = neg ? ni + 1 : offset;
            stack[++top] = far_i;
            stack[++top] = near_i;
        }
    }
    return h;
}

// ---- hit decoding -----------------------------------------------------------------------------
struct Surface {
    int kind;       // 0 triangle, 1 sphere, 2 plane
    int index;      // triangle / sphere / plane index
    int instance;   // placement (triangles)
    int material;
    int toff, tcount;
};

__device__ Surface surface_of(const DevScene &S, const Hit &h) {
    Surface s;
    if (h.prim <= -2) {
        s.kind = 2;
        s.index = -2 - h.prim;
        s.instance = -1;
        const rt_plane &p = S.planes[s.index];
        s.material = p.material;
        s.toff = p.tex_offset;
        s.tcount = p.tex_count;
        return s;
    }
    const float4 p2 = S.prims[3 * h.prim + 2];
    const int inst = __float_as_int(p2.y);
    if (inst >= 0) {
        s.kind = 0;
        s.index = __float_as_int(p2.z);
        s.instance = inst;
        const rt_instance &in = S.instances[inst];
        s.material = in.material;
        s.toff = in.tex_offset;
        s.tcount = in.tex_count;
    } else {
        s.kind = 1;
        s.index = -1 - inst;
        s.instance = -1;
        const rt_sphere &sp = S.spheres[s.index];
        s.material = sp.material;
        s.toff = sp.tex_offset;
        s.tcount = sp.tex_count;
    }
    return s;
}

// get_normal: triangle src/shapes.cpp:152-197, sphere :390-405, plane :504
__device__ V normal_of(const DevScene &S, const Surface &s, V p) {
    if (s.kind == 0) {
        const rt_triangle &t = S.triangles[s.index];
        const rt_instance &in = S.instances[s.instance];
        if (!t.smooth) return normalize(xdir(in.normal, mk(S.flat_normals + 3 * s.index)));
        V po = xpoint(in.inv_model, p);
        V c1 = mk(t.c1);
        V v0 = sub(mk(t.c2), c1), v1 = sub(mk(t.c3), c1), v2 = sub(po, c1);
        float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
        const float denom = d00 * d11 - d01 * d01;
        float v = rt_div(d11 * d20 - d01 * d21, denom);
        float w = rt_div(d00 * d21 - d01 * d20, denom);
        float u = 1.0f - v - w;
        V n = add(add(scl(mk(t.n1), u), scl(mk(t.n2), v)), scl(mk(t.n3), w));
        n = normalize(n);
        return normalize(xdir(in.normal, n));
    }
    if (s.kind == 1) {
        const rt_sphere &sp = S.spheres[s.index];
        V po = xpoint(sp.inv_model, p);
        return normalize(xdir(sp.normal, sub(po, mk(sp.center))));
    }
    return mk(S.planes[s.index].normal);
}

// calculate_uv: triangle src/shapes.cpp:87-122, sphere :449-465, plane :529-548
__device__ void uv_of(const DevScene &S, const Surface &s, V p, float &u, float &v) {
    if (s.kind == 0) {
        const rt_triangle &t = S.triangles[s.index];
        const rt_instance &in = S.instances[s.instance];
        V local = xpoint(in.inv_model, p);
        V c1 = mk(t.c1);
        V e1 = sub(mk(t.c2), c1), e2 = sub(mk(t.c3), c1), vp = sub(local, c1);
        float d00 = dot(e1, e1), d01 = dot(e1, e2), d11 = dot(e2, e2), d20 = dot(vp, e1), d21 = dot(vp, e2);
        float denom = d00 * d11 - d01 * d01;
        if (absf(denom) < 1e-10f) {
            u = 0.0f;
            v = 0.0f;
            return;
        }
        float inv = rt_div(1.0f, denom);
        float beta = (d11 * d20 - d01 * d21) * inv;
        float gamma = (d00 * d21 - d01 * d20) * inv;
        float alpha = 1.0f - beta - gamma;
        u = alpha * t.tex_u[0] + beta * t.tex_u[1] + gamma * t.tex_u[2];
        v = alpha * t.tex_v[0] + beta * t.tex_v[1] + gamma * t.tex_v[2];
        return;
    }
    if (s.kind == 1) {
        const rt_sphere &sp = S.spheres[s.index];
        V q = normalize(sub(xpoint(sp.inv_model, p), mk(sp.center)));
        float phi = rt_atan2(q.z, q.x);
        float theta = rt_asin(q.y);
        const float PI = 3.14159265359f;
        u = 1.0f - rt_div(phi + PI, 2.0f * PI);
        v = 1.0f - rt_div(theta + PI / 2.0f, PI);
        return;
    }
    const rt_plane &pl = S.planes[s.index];
    V n = mk(pl.normal);
    V helper = absf(pl.raw_normal_x) > 0.9f ? mk(0, 1, 0) : mk(1, 0, 0);
    V tangent = normalize(cross(n, helper));
    V bitangent = cross(n, tangent);
    V to = sub(p, mk(pl.point));
    u = dot(to, tangent);
    v = dot(to, bitangent);
}

// get_tangent: triangle src/shapes.cpp:281-316, sphere :467-494, plane :550-562
__device__ V tangent_of(const DevScene &S, const Surface &s, V p) {
    if (s.kind == 0) {
        const rt_triangle &t = S.triangles[s.index];
        const rt_instance &in = S.instances[s.instance];
        V c1 = mk(t.c1);
        V e1 = sub(mk(t.c2), c1), e2 = sub(mk(t.c3), c1);
        float du1 = t.tex_u[1] - t.tex_u[0], dv1 = t.tex_v[1] - t.tex_v[0];
        float du2 = t.tex_u[2] - t.tex_u[0], dv2 = t.tex_v[2] - t.tex_v[0];
        float det = (du1 * dv2 - du2 * dv1);
        V lt;
        if (absf(det) < 1e-8f) {
            lt = e1;
        } else {
            float f = rt_div(1.0f, det);
            lt = scl(sub(scl(e1, dv2), scl(e2, dv1)), f);
        }
        return normalize(xdir(in.model, lt));
    }
    if (s.kind == 1) {
        // the reference's object-space sphere tangent is always (1, 0, 0): src/shapes.cpp:473-487 reads scalar
        // copies of a difference vector that only exists in SIMD lanes
        return normalize(xdir(S.spheres[s.index].model, mk(1, 0, 0)));
    }
    const rt_plane &pl = S.planes[s.index];
    V helper = absf(pl.raw_normal_x) > 0.9f ? mk(0, 1, 0) : mk(1, 0, 0);
    return normalize(cross(mk(pl.normal), helper));
}

// ---- textures (src/texture.cpp) ------------------------------------------------------------------
__device__ bool image_sample(const DevScene &S, const rt_image &im, bool bilinear, float u, float v, V &color, Counters &c) {
    if (im.width <= 0 || im.height <= 0) return false;
    const int W = im.width, H = im.height;
    u -= floorf(u);
    v -= floorf(v);
    if (!bilinear) {   // image::sample_nearest :64-93
        int x = (int)(u * W), y = (int)(v * H);
        if (x >= W) x = W - 1;
        if (y >= H) y = H - 1;
        ++c.texels;
        if (im.is_hdr) {
            const float *p = S.texels_f32 + (size_t)im.offset + ((size_t)y * W + x) * 4;
            color = mk(p[0], p[1], p[2]);
        } else {
            const uint8_t *p = S.texels_u8 + (size_t)im.offset + ((size_t)y * W + x) * 3;
            color = mk(p[0] * (1.0f / 255.0f), p[1] * (1.0f / 255.0f), p[2] * (1.0f / 255.0f));
        }
        return true;
    }
    // image::sample_bilinear :95-166
    float px = u * W - 0.5f, py = v * H - 0.5f;
    int x = (int)floorf(px), y = (int)floorf(py);
    float ur = px - x, vr = py - y;
    float uo = 1.0f - ur, vo = 1.0f - vr;
    int x0 = (x % W + W) % W, x1 = ((x + 1) % W + W) % W;
    int y0 = (y % H + H) % H, y1 = ((y + 1) % H + H) % H;
    c.texels += 4;
    float out[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float v00, v10, v01, v11;
        if (im.is_hdr) {
            const float *b = S.texels_f32 + (size_t)im.offset;
            v00 = b[((size_t)y0 * W + x0) * 4 + ch];
            v10 = b[((size_t)y0 * W + x1) * 4 + ch];
            v01 = b[((size_t)y1 * W + x0) * 4 + ch];
            v11 = b[((size_t)y1 * W + x1) * 4 + ch];
        } else {
            const uint8_t *b = S.texels_u8 + (size_t)im.offset;
            v00 = b[((size_t)y0 * W + x0) * 3 + ch];
            v10 = b[((size_t)y0 * W + x1) * 3 + ch];
            v01 = b[((size_t)y1 * W + x0) * 3 + ch];
            v11 = b[((size_t)y1 * W + x1) * 3 + ch];
        }
        float top = v00 * uo + v10 * ur;
        float bot = v01 * uo + v11 * ur;
        out[ch] = top * vo + bot * vr;
    }
    if (im.is_hdr)
        color = mk(out[0], out[1], out[2]);
    else
        color = mk(out[0] * (1.0f / 255.0f), out[1] * (1.0f / 255.0f), out[2] * (1.0f / 255.0f));
    return true;
}

__constant__ uint8_t PERLIN_P[256] = {   // Ken Perlin's reference permutation (src/texture.cpp:238)
    151, 160, 137, 91,  90,  15,  131, 13,  201, 95,  96,  53,  194, 233, 7,   225, 140, 36,  103, 30,  69,  142, 8,   99,  37,  240,
    21,  10,  23,  190, 6,   148, 247, 120, 234, 75,  0,   26,  197, 62,  94,  252, 219, 203, 117, 35,  11,  32,  57,  177, 33,  88,
    237, 149, 56,  87,  174, 20,  125, 136, 171, 168, 68,  175, 74,  165, 71,  134, 139, 48,  27,  166, 77,  146, 158, 231, 83,  111,
    229, 122, 60,  211, 133, 230, 220, 105, 92,  41,  55,  46,  245, 40,  244, 102, 143, 54,  65,  25,  63,  161, 1,   216, 80,  73,
    209, 76,  132, 187, 208, 89,  18,  169, 200, 196, 135, 130, 116, 188, 159, 86,  164, 100, 109, 198, 173, 186, 3,   64,  52,  217,
    226, 250, 124, 123, 5,   202, 38,  147, 118, 126, 255, 82,  85,  212, 207, 206, 59,  227, 47,  16,  58,  17,  182, 189, 28,  42,
    223, 183, 170, 213, 119, 248, 152, 2,   44,  154, 163, 70,  221, 153, 101, 155, 167, 43,  172, 9,   129, 22,  39,  253, 19,  98,
    108, 110, 79,  113, 224, 232, 178, 185, 112, 104, 218, 246, 97,  228, 251, 34,  242, 193, 238, 210, 144, 12,  191, 179, 162, 241,
    81,  51,  145, 235, 249, 14,  239, 107, 49,  192, 214, 31,  181, 199, 106, 157, 184, 84,  204, 176, 115, 121, 50,  45,  127, 4,
    150, 254, 138, 236, 205, 93,  222, 114, 67,  29,  24,  72,  243, 141, 128, 195, 78,  66,  215, 61,  156, 180};

__device__ __forceinline__ float pfade(float t) { return t * t * t * (t * (t * 6 - 15) + 10); }
__device__ __forceinline__ float plerp(float t, float a, float b) { return a + t * (b - a); }
__device__ __forceinline__ float pgrad(int hash, float x, float y, float z) {
    int h = hash & 15;
    float u = h < 8 ? x : y;
    float v = h < 4 ? y : (h == 12 || h == 14 ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}
__device__ float perlin_noise(float x, float y, float z) {   // Perlin::noise :204-223
    auto p = [](int i) -> int { return PERLIN_P[i & 255]; };
    int X = (int)floorf(x) & 255, Y = (int)floorf(y) & 255, Z = (int)floorf(z) & 255;
    x -= floorf(x);
    y -= floorf(y);
    z -= floorf(z);
    float u = pfade(x), v = pfade(y), w = pfade(z);
    int A = p(X) + Y, AA = p(A) + Z, AB = p(A + 1) + Z;
    int B = p(X + 1) + Y, BA = p(B) + Z, BB = p(B + 1) + Z;
    return plerp(w,
                 plerp(v, plerp(u, pgrad(p(AA), x, y, z), pgrad(p(BA), x - 1, y, z)),
                       plerp(u, pgrad(p(AB), x, y - 1, z), pgrad(p(BB), x - 1, y - 1, z))),
                 plerp(v, plerp(u, pgrad(p(AA + 1), x, y, z - 1), pgrad(p(BA + 1), x - 1, y, z - 1)),
                       plerp(u, pgrad(p(AB + 1), x, y - 1, z - 1), pgrad(p(BB + 1), x - 1, y - 1, z - 1))));
}
__device__ float perlin(V h, const rt_texture &t) {   // perlin() :241-274
    float s = 0.0f;
    float x = h.x * t.noise_scale, y = h.y * t.noise_scale, z = h.z * t.noise_scale;
    float weight = 1.0f, max_amp = 0.0f;
    int K = (int)t.num_octaves;
    if (K < 1) K = 1;
    for (int k = 0; k < K; ++k) {
        s += weight * perlin_noise(x, y, z);
        max_amp += weight;
        x *= 2.0f;
        y *= 2.0f;
        weight *= 0.5f;
    }
    float sn = rt_div(s, max_amp);
    if (t.noise_conversion == RT_NOISE_ABSVAL) return absf(sn);
    return (sn + 1.0f) * 0.5f;
}
// texture::sample :168-197
__device__ void tex_sample(const DevScene &S, const rt_texture &t, float u, float v, V h, V &color, Counters &c) {
    if (t.image >= 0) {
        image_sample(S, S.images[t.image], t.interp != RT_INTERP_NEAREST, u, v, color, c);
    } else if (t.kind == RT_TEX_CHECKER) {
        int x = (int)floorf((u + t.offset) * t.scale);
        int y = (int)floorf((v + t.offset) * t.scale);
        color = ((x + y) % 2 != 0) ? mk(t.black) : mk(t.white);
    } else {
        float x = perlin(h, t);
        color = mk(x, x, x);
    }
}

// get_latlong_uv / get_probe_uv, src/ray_trace.cpp:7-39
__device__ void latlong_uv(V d, float &u, float &v) {
    const float pi = 3.14159265359f;
    u = (1.0f + rt_div(rt_atan2(d.x, -d.z), pi)) * 0.5f;
    v = rt_div(rt_acos(d.y), pi);
}
__device__ void probe_uv(V d, float &u, float &v) {
    const float pi = 3.14159265359f;
    float len_sq = d.x * d.x + d.y * d.y;
    float r = 0.0f;
    if (len_sq > 1e-6f) r = rt_div((1.0f / pi) * rt_acos(-d.z), rt_sqrt(len_sq));
    u = (r * d.x + 1.0f) * 0.5f;
    v = (-r * d.y + 1.0f) * 0.5f;
}
// miss / depth-limit colour (src/ray_trace.cpp:649-679, 691-721, 887-903)
__device__ V background(const DevScene &S, const rt_camera_desc &cam, V dir, bool whitted, float pixelu, float pixelv, Counters &c) {
    if (S.bg_tex >= 0) {
        const rt_texture &bg = S.textures[S.bg_tex];
        if (bg.image >= 0) {
            V color = mk(0, 0, 0);
            if (!whitted || cam.is_hdr) {
                float u, v;
                if (S.is_probe)
                    probe_uv(dir, u, v);
                else
                    latlong_uv(dir, u, v);
                tex_sample(S, bg, u, v, mk(0, 0, 0), color, c);
            } else {
                tex_sample(S, bg, pixelu, pixelv, mk(pixelu, pixelv, 0), color, c);
            }
            return color;
        }
    }
    return mk(S.bg);
}

// ---- directions (src/ray_trace.cpp:41-105, 408-492) ------------------------------------------------
__device__ void sample_hemisphere(bool cosine, float r1, float r2, V normal, float helper_x, V &dir, float &pdf) {
    float phi = 2.0f * 3.14159265359f * r1;
    float sin_theta, cos_theta;
    if (cosine) {
        sin_theta = rt_sqrt(r2);
        cos_theta = rt_sqrt(1.0f - r2);
    } else {
        sin_theta = rt_sqrt(1.0f - r2 * r2);
        cos_theta = r2;
    }
    float x = sin_theta * rt_cos(phi), y = sin_theta * rt_sin(phi), z = cos_theta;
    V w = normal;
    V helper = absf(helper_x) > 0.9f ? mk(0, 1, 0) : mk(1, 0, 0);
    V u = normalize(cross(helper, w));
    V v = normalize(cross(w, u));
    dir = normalize(add(add(scl(u, x), scl(v, y)), scl(w, z)));
    pdf = cosine ? rt_div(cos_theta, 3.14159265359f) : 1.0f / (2.0f * 3.14159265359f);
}
__device__ void add_roughness(V &R, float roughness, float r_xi1, float r_xi2) {
    float ax = absf(R.x), ay = absf(R.y), az = absf(R.z);
    V rp;
    if (ax <= ay && ax <= az)
        rp = mk(1.0f, R.y, R.z);
    else if (ay <= az)
        rp = mk(R.x, 1.0f, R.z);
    else
        rp = mk(R.x, R.y, 1.0f);
    V u = normalize(cross(R, rp));
    V v = cross(R, u);
    float xi1 = r_xi1 - 0.5f, xi2 = r_xi2 - 0.5f;
    V off = scl(add(scl(u, xi1), scl(v, xi2)), roughness);
    R = add(R, off);
}
__device__ V reflect_dir(V N, V I, float roughness, uint32_t key) {
    float d = dot(I, N);
    V R = add(I, scl(N, d * -2.0f));
    if (roughness > 0.0f) add_roughness(R, roughness, rnd(key, DIM_REFLECT_XI1), rnd(key, DIM_REFLECT_XI2));
    return normalize(R);
}
__device__ bool refract_dir(V N, V I, float n1, float n2, float roughness, uint32_t key, V &T) {
    float cosi = -dot(I, N);
    float eta = rt_div(n1, n2);
    float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (k < 0.0f) return false;
    T = add(scl(I, eta), scl(N, (eta * cosi - rt_sqrt(k))));
    if (roughness > 0.0f) add_roughness(T, roughness, rnd(key, DIM_REFRACT_XI1), rnd(key, DIM_REFRACT_XI2));
    T = normalize(T);
    return true;
}
// exact conductor Fresnel reflectance, src/ray_trace.cpp:751-786
__device__ float conductor_fresnel(float cos_theta, float n, float k) {
    float c = cos_theta, c2 = c * c, s2 = 1.0f - c2;
    float n2 = n * n, k2 = k * k, n2_k2 = n2 - k2, n2k2_4 = 4.0f * n2 * k2;
    float a2_b2 = rt_sqrt((n2_k2 - s2) * (n2_k2 - s2) + n2k2_4);
    float a2 = 0.5f * (a2_b2 + n2_k2 - s2);
    float a = rt_sqrt(a2);
    float b2 = a2_b2 - a2;
    float b = rt_sqrt(b2);
    float rs_num = a2_b2 + c2 - (2.0f * a * c);
    float rs_den = a2_b2 + c2 + (2.0f * a * c);
    float rs = (rs_den == 0.0f) ? 1.0f : rt_div(rs_num, rs_den);
    float ac = a * c, bc = b * c;
    float t1n = ac - s2, t1d = ac + s2;
    float rp_num = (t1n * t1n) + (bc * bc);
    float rp_den = (t1d * t1d) + (bc * bc);
    float rp = (rp_den == 0.0f) ? 1.0f : rs * rt_div(rp_num, rp_den);
    float F = 0.5f * (rs + rp);
    return F < 0.0f ? 0.0f : (F > 1.0f ? 1.0f : F);
}

// ---- normal / bump mapping (src/ray_trace.cpp:494-644) ------------------------------------------------
__device__ void apply_normal_map(const DevScene &S, const Surface &s, V p, V &normal, Counters &c) {
    for (int k = 0; k < s.tcount; ++k) {
        const rt_texture &tex = S.textures[S.texture_refs[s.toff + k]];
        if (tex.decal_mode != RT_DECAL_REPLACE_NORMAL) continue;
        float u, v;
        uv_of(S, s, p, u, v);
        V raw = mk(0.5f, 0.5f, 1.0f);
        tex_sample(S, tex, u, v, p, raw, c);
        V ts = sub(scl(raw, 2.0f), mk(1.0f, 1.0f, 1.0f));
        V T = tangent_of(S, s, p);
        float ndott = dot(normal, T);
        T = normalize(sub(T, scl(normal, ndott)));
        V B = cross(normal, T);
        V nn = add(add(scl(T, ts.x), scl(B, ts.y)), scl(normal, ts.z));
        normal = normalize(nn);
        return;
    }
}
__device__ void apply_bump_map(const DevScene &S, const Surface &s, V p, V &normal, Counters &c) {
    for (int k = 0; k < s.tcount; ++k) {
        const rt_texture &tex = S.textures[S.texture_refs[s.toff + k]];
        if (tex.decal_mode != RT_DECAL_BUMP_NORMAL) continue;
        if (tex.image >= 0) {
            const rt_image &im = S.images[tex.image];
            float u, v;
            uv_of(S, s, p, u, v);
            float du = rt_div(1.0f, (float)im.width), dv = rt_div(1.0f, (float)im.height);
            V hc = mk(0, 0, 0), hu = mk(0, 0, 0), hv = mk(0, 0, 0);
            tex_sample(S, tex, u, v, p, hc, c);
            tex_sample(S, tex, u + du, v, p, hu, c);
            tex_sample(S, tex, u, v + dv, p, hv, c);
            float dh_du = (hu.x - hc.x) * tex.bump_factor;
            float dh_dv = (hv.x - hc.x) * tex.bump_factor;
            V T = tangent_of(S, s, p);
            float ndott = dot(normal, T);
            T = normalize(sub(T, scl(normal, ndott)));
            V B = cross(normal, T);
            V nn = sub(sub(normal, scl(T, dh_du)), scl(B, dh_dv));
            normal = normalize(nn);
        } else {
            const float eps = 0.01f;
            const float inv_eps = 1.0f / eps;
            V hc, hx, hy, hz;
            tex_sample(S, tex, 0, 0, p, hc, c);
            tex_sample(S, tex, 0, 0, add(p, mk(eps, 0, 0)), hx, c);
            tex_sample(S, tex, 0, 0, add(p, mk(0, eps, 0)), hy, c);
            tex_sample(S, tex, 0, 0, add(p, mk(0, 0, eps)), hz, c);
            V g = mk((hx.x - hc.x) * inv_eps, (hy.x - hc.x) * inv_eps, (hz.x - hc.x) * inv_eps);
            float gdn = dot(g, normal);
            V surf = sub(g, scl(normal, gdn));
            normal = normalize(sub(normal, scl(surf, tex.bump_factor)));
        }
        return;
    }
}

// resolve_textures, src/ray_trace.cpp:107-152.  kd_rr: the diffuse reflectance Russian roulette reads
// (the reference's blend_kd update does not reach the scalar copies that roulette looks at).
__device__ void resolve_textures(const DevScene &S, const Surface &s, V p, const rt_material &m, V &kd, V &ks, V &kd_rr,
                                 bool &replace_all, V &replace_color, Counters &c) {
    kd = mk(m.diffuse);
    ks = mk(m.specular);
    kd_rr = kd;
    replace_all = false;
    for (int k = 0; k < s.tcount; ++k) {
        const rt_texture &tex = S.textures[S.texture_refs[s.toff + k]];
        float u, v;
        if (tex.decal_mode == RT_DECAL_REPLACE_KD) {
            uv_of(S, s, p, u, v);
            tex_sample(S, tex, u, v, p, kd, c);
            kd_rr = kd;
        } else if (tex.decal_mode == RT_DECAL_BLEND_KD) {
            uv_of(S, s, p, u, v);
            V tmp = mk(0, 0, 0);
            tex_sample(S, tex, u, v, p, tmp, c);
            kd = scl(add(mk(m.diffuse), tmp), 0.5f);
        } else if (tex.decal_mode == RT_DECAL_REPLACE_KS) {
            uv_of(S, s, p, u, v);
            tex_sample(S, tex, u, v, p, ks, c);
        } else if (tex.decal_mode == RT_DECAL_REPLACE_ALL) {
            uv_of(S, s, p, u, v);
            replace_color = mk(0, 0, 0);
            tex_sample(S, tex, u, v, p, replace_color, c);
            replace_all = true;
            return;
        }
    }
}

// ---- lights (Light::get_sample, src/lights.cpp) ------------------------------------------------------
__device__ void light_sample(const DevScene &S, const rt_light &L, V p, V normal, float ru, float rv, V &radiance, V &ldir,
                             float &dist, Counters &c) {
    const float pi = 3.14159265359f;
    radiance = mk(0, 0, 0);
    switch (L.type) {
        case RT_LIGHT_POINT: {
            V du = sub(mk(L.position), p);
            float d2 = dot(du, du);
            dist = rt_sqrt(d2);
            if (d2 < 1e-6f) {
                ldir = du;
                return;
            }
            ldir = scl(du, rt_div(1.0f, dist));
            radiance = scl(mk(L.color), rt_div(1.0f, d2));
            return;
        }
        case RT_LIGHT_AREA: {
            float uo = (ru - 0.5f) * L.size, vo = (rv - 0.5f) * L.size;
            V sp = add(add(mk(L.position), scl(mk(L.u), uo)), scl(mk(L.v), vo));
            V du = sub(sp, p);
            float d2 = dot(du, du);
            dist = rt_sqrt(d2);
            ldir = scl(du, rt_div(1.0f, dist));
            float cos_l = dot(mk(L.normal), scl(ldir, -1.0f));
            if (cos_l <= 0.0f) return;
            float area = L.size * L.size;
            radiance = scl(mk(L.color), rt_div(area * cos_l, d2));
            return;
        }
        case RT_LIGHT_DIRECTIONAL: {
            dist = 3.402823466e+38f;
            ldir = scl(mk(L.direction), -1.0f);
            radiance = mk(L.color);
            return;
        }
        case RT_LIGHT_SPOT: {
            V du = sub(mk(L.position), p);
            float d2 = dot(du, du);
            dist = rt_sqrt(d2);
            if (d2 < 1e-6f) {
                ldir = du;
                return;
            }
            ldir = scl(du, rt_div(1.0f, dist));
            float cos_alpha = dot(scl(ldir, -1.0f), mk(L.direction));
            float spot = 0.0f;
            if (cos_alpha > L.cos_falloff) {
                spot = 1.0f;
            } else if (cos_alpha > L.cos_coverage) {
                float ratio = rt_div(cos_alpha - L.cos_coverage, L.cos_falloff - L.cos_coverage);
                spot = ratio * ratio * ratio * ratio;
            }
            radiance = scl(mk(L.color), rt_div(spot, d2));
            return;
        }
        case RT_LIGHT_ENV: {
            dist = 3.402823466e+38f;
            float phi = 2.0f * pi * ru, cos_t, sin_t, pdf;
            if (L.cosine_sampling) {
                cos_t = rt_sqrt(rv);
                sin_t = rt_sqrt(1.0f - cos_t * cos_t);
                pdf = rt_div(cos_t, pi);
            } else {
                cos_t = rv;
                sin_t = rt_sqrt(1.0f - cos_t * cos_t);
                pdf = 1.0f / (2.0f * pi);
            }
            float lx = sin_t * rt_cos(phi), ly = sin_t * rt_sin(phi), lz = cos_t;
            V w = normal;
            V up = absf(w.y) < 0.9f ? mk(0, 1, 0) : mk(0, 0, 1);
            V u = normalize(cross(up, w));
            V vb = normalize(cross(w, u));
            ldir = normalize(add(add(scl(u, lx), scl(vb, ly)), scl(w, lz)));
            float uu, vv;
            if (L.is_probe)
                probe_uv(ldir, uu, vv);
            else
                latlong_uv(ldir, uu, vv);
            if (L.image >= 0) image_sample(S, S.images[L.image], true, uu, vv, radiance, c);
            if (pdf < 1e-6f) pdf = 1e-6f;
            radiance = scl(radiance, rt_div(1.0f, pdf));
            return;
        }
        case RT_LIGHT_TRIANGLE: {
            float sr = rt_sqrt(ru);
            float u = 1.0f - sr, v = rv * sr, w = 1.0f - u - v;
            V sp = add(add(scl(mk(L.v0), w), scl(mk(L.v1), u)), scl(mk(L.v2), v));
            V du = sub(sp, p);
            float d2 = dot(du, du);
            dist = rt_sqrt(d2);
            ldir = scl(du, rt_div(1.0f, dist));
            float cos_l = absf(dot(mk(L.normal), scl(ldir, -1.0f)));
            radiance = scl(mk(L.color), rt_div(L.area * cos_l, d2));
            return;
        }
        case RT_LIGHT_SPHERE: {
            const rt_sphere &s = S.spheres[L.sphere];
            float phi = 2.0f * 3.14159265359f * ru;
            float cos_t = 1.0f - 2.0f * rv;
            float sin_t = rt_sqrt(max_std(0.0f, 1.0f - cos_t * cos_t));
            V unit = mk(sin_t * rt_cos(phi), sin_t * rt_sin(phi), cos_t);
            // the reference transforms the scalar copies of the sampled point, which are still zero
            // (src/lights.cpp:418-422): the position used is model * (0,0,0,1)
            V sp = xpoint(s.model, mk(0, 0, 0));
            V nl = normalize(xdir(s.normal, unit));
            V du = sub(sp, p);
            float d2 = dot(du, du);
            dist = rt_sqrt(d2);
            ldir = scl(du, rt_div(1.0f, dist));
            float cos_l = dot(nl, scl(ldir, -1.0f));
            if (cos_l <= 0.0f) return;
            radiance = scl(mk(L.color), rt_div(L.area * cos_l, d2));
            return;
        }
    }
    ldir = mk(0, 0, 0);
    dist = 0.f;
}

// calculate_color, src/ray_trace.cpp:154-406 (the last-blocker cache :158-159,207-226 is not used).
template <bool COUNT>
__device__ V calculate_color(const DevScene &S, V normal, const rt_material &m, const Surface &s, V p, V ray_origin, float time,
                             bool is_hdr, bool include_ambient, uint32_t key, Counters &c) {
    V kd, ks, kd_rr, rep;
    bool replace_all;
    resolve_textures(S, s, p, m, kd, ks, kd_rr, replace_all, rep, c);
    if (replace_all) return rep;
    V color = mk(0, 0, 0);
    if (include_ambient) color = add(color, mul(mk(m.ambient), mk(S.ambient)));
    const V view = normalize(sub(ray_origin, p));
    const float seps = S.seps;
    uint32_t k = 0;
    for (int li = 0; li < S.n_lights; ++li) {
        const rt_light &L = S.lights[li];
        const int ns = L.samples;
        const float weight = rt_div(1.0f, (float)ns);
        for (int si = 0; si < ns; ++si, ++k) {
            float r1 = rnd(key, DIM_LIGHT + 2 * k), r2 = rnd(key, DIM_LIGHT + 2 * k + 1);
            V rad, ldir;
            float dist;
            light_sample(S, L, p, normal, r1, r2, rad, ldir, dist, c);
            if (rad.x == 0 && rad.y == 0 && rad.z == 0) continue;
            V so = add(p, scl(ldir, seps));
            ++c.shadow;
            Hit sh = traverse<true, COUNT>(S, so, ldir, time, false, seps, dist, c);
            if (sh.prim != -1 && sh.t < dist - seps) continue;
            float ndotl = max_std(dot(normal, ldir), 0.0f);
            V hv = normalize(add(ldir, view));
            V fd = mk(0, 0, 0), fs = mk(0, 0, 0);
            const float p_exp = m.phong_exponent;
            const float PI = 3.14159265359f;
            switch (m.brdf) {
                case RT_BRDF_ORIGINAL_BLINN_PHONG: {
                    fd = scl(kd, ndotl);
                    float ndoth = max_std(dot(normal, hv), 0.0f);
                    fs = scl(ks, rt_pow(ndoth, p_exp));
                    break;
                }
                case RT_BRDF_ORIGINAL_PHONG: {
                    fd = scl(kd, ndotl);
                    V R = reflect_dir(normal, scl(ldir, -1.0f), 0.f, 0);
                    float rdoth = max_std(dot(R, view), 0.0f);
                    fs = scl(ks, rt_pow(rdoth, p_exp));
                    break;
                }
                case RT_BRDF_MODIFIED_BLINN_PHONG: {
                    float ndoth = max_std(dot(normal, hv), 0.0f);
                    if (m.normalized) {
                        fd = scl(kd, 1.0f / PI * ndotl);
                        float norm = rt_div(p_exp + 8.0f, 8.0f * PI);
                        fs = scl(ks, norm * rt_pow(ndoth, p_exp) * ndotl);
                    } else {
                        fd = scl(kd, ndotl);
                        fs = scl(ks, rt_pow(ndoth, p_exp) * ndotl);
                    }
                    break;
                }
                case RT_BRDF_MODIFIED_PHONG: {
                    V R = reflect_dir(normal, scl(ldir, -1.0f), 0.f, 0);
                    float rdoth = max_std(dot(R, view), 0.0f);
                    if (m.normalized) {
                        fd = scl(kd, 1.0f / PI * ndotl);
                        float norm = rt_div(p_exp + 2.0f, 2.0f * PI);
                        fs = scl(ks, norm * rt_pow(rdoth, p_exp) * ndotl);
                    } else {
                        fd = scl(kd, ndotl);
                        fs = scl(ks, rt_pow(rdoth, p_exp) * ndotl);
                    }
                    break;
                }
                case RT_BRDF_TORRANCE_SPARROW: {
                    float ndoth = max_std(dot(normal, hv), 0.0f);
                    float ndotv = max_std(dot(normal, view), 0.0f);
                    float vdoth = max_std(dot(view, hv), 0.0f);
                    float D = rt_div(p_exp + 2.0f, 2.0f * PI) * rt_pow(ndoth, p_exp);
                    float f_term = rt_pow(1.0f - vdoth, 5.0f);
                    V one = mk(1, 1, 1);
                    V F = add(ks, scl(sub(one, ks), f_term));
                    float G = 1.0f;
                    if (vdoth > 1e-6f) {
                        float g1 = rt_div(2.0f * ndoth * ndotv, vdoth);
                        float g2 = rt_div(2.0f * ndoth * ndotl, vdoth);
                        G = min_std(1.0f, min_std(g1, g2));
                    }
                    if (ndotv > 1e-6f) {
                        float denom = 4.0f * ndotv;
                        fs = scl(F, rt_div(D * G, denom));
                    }
                    if (m.kdfresnel)
                        fd = mul(scl(kd, ndotl), sub(one, F));
                    else
                        fd = scl(kd, 1.0f / PI * ndotl);
                    break;
                }
            }
            fd = scl(mul(fd, rad), weight);
            fs = scl(mul(fs, rad), weight);
            color = add(color, fd);
            color = add(color, fs);
        }
    }
    if (!is_hdr) color = scl(color, 0.004f);
    return color;
}

// ---- camera (camera::get_samples, src/camera.cpp:117-224) --------------------------------------------
__device__ void camera_sample(const rt_camera_desc &cam, int i, int j, int s, int grid_n, uint32_t key, V &o, V &dir, float &time) {
    V pos = mk(cam.position), cu = mk(cam.u), cv = mk(cam.v), cw = mk(cam.w), center = mk(cam.center);
    if (cam.num_samples == 1) {
        float su = cam.near_left + rt_div((cam.near_right - cam.near_left) * (i + 0.5f), (float)cam.resx);
        float sv = cam.near_bottom + rt_div((cam.near_top - cam.near_bottom) * (j + 0.5f), (float)cam.resy);
        V target = add(add(center, scl(cu, su)), scl(cv, sv));
        dir = normalize(sub(target, pos));
        o = pos;
        time = rnd(key, DIM_CAM_TIME);
        return;
    }
    int sy = s / grid_n, sx = s % grid_n;
    float r1 = rnd(key, DIM_CAM_R1), r2 = rnd(key, DIM_CAM_R2);
    float sub_u = rt_div(sx + r1, (float)grid_n);
    float sub_v = rt_div(sy + r2, (float)grid_n);
    float su = cam.near_left + rt_div((cam.near_right - cam.near_left) * (i + sub_u), (float)cam.resx);
    float sv = cam.near_bottom + rt_div((cam.near_top - cam.near_bottom) * (j + sub_v), (float)cam.resy);
    V target = add(add(center, scl(cu, su)), scl(cv, sv));
    dir = normalize(sub(target, pos));
    o = pos;
    if (cam.aperture_size > 0.0f && cam.focus_distance > 0.0f) {
        float cos_theta = -dot(dir, cw);
        if (absf(cos_theta) < 1e-6f) cos_theta = 1e-6f;
        float t = rt_div(cam.focus_distance, cos_theta);
        V focal = add(pos, scl(dir, t));
        float r3 = rnd(key, DIM_CAM_R3), r4 = rnd(key, DIM_CAM_R4);
        float lens_x = (r3 - 0.5f) * cam.aperture_size;
        float lens_y = (r4 - 0.5f) * cam.aperture_size;
        V lens = add(scl(cu, lens_x), scl(cv, lens_y));
        o = add(pos, lens);
        dir = normalize(sub(focal, o));
    }
    time = rnd(key, DIM_CAM_TIME);
}

// ---- pending rays of one camera sample -----------------------------------------------------------------
struct PathItem {
    V o, d, thr;
    uint32_t key;
    int depth_flags;   // depth | culling << 16 | diffuse bounce << 17
};

struct LaunchParams {
    int tiles_x, tiles_y;   // image size in 32x32 tiles
    int rank, world;        // tile ownership: tile k belongs to rank k % world
    int n_owned;            // tiles of this rank
    int grid_n;             // floor(sqrt(num_samples))
    int n_samples;          // grid_n^2 (or 1)
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One lane = one pixel.  process_pixel (src/main.cpp:451-484) + ray_tracer::trace (src/ray_trace.cpp:1141-1172)
// with trace_rec (:646-869) / path_trace (:871-1139) flattened into a depth-first walk over PathItems.
template <bool COUNT>
__global__ void __launch_bounds__(BLOCK_EDGE *BLOCK_EDGE)
    render_kernel(DevScene S, rt_camera_desc cam, LaunchParams lp, float *__restrict__ out, unsigned long long *__restrict__ stats) {
    // workgroup -> (owned tile, 16x16 quadrant); wave -> 8x8 sub-tile; lane -> pixel
    const int owned = blockIdx.x >> 2, quad = blockIdx.x & 3;
    const int tile = owned * lp.world + lp.rank;
    const int tx = tile % lp.tiles_x, ty = tile / lp.tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = tx * TILE + (quad & 1) * BLOCK_EDGE + (wave & 1) * 8 + (lane & 7);
    const int j = ty * TILE + (quad >> 1) * BLOCK_EDGE + (wave >> 1) * 8 + (lane >> 3);
    Counters c;
    memset(&c, 0, sizeof c);
    if (i < cam.resx && j < cam.resy) {
        const float pixelv = 1.0f - rt_div((float)j, (float)(cam.resy - 1));
        const float pixelu = rt_div((float)i, (float)(cam.resx - 1));
        const uint32_t pixel_index = (uint32_t)j * (uint32_t)cam.resx + (uint32_t)i;
        const bool is_hdr = cam.is_hdr != 0;
        const bool pt = cam.path_tracing != 0;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        PathItem stack[PATH_STACK];
        for (int s = 0; s < lp.n_samples; ++s) {
            const uint32_t skey = sample_key(cam.seed, pixel_index, (uint32_t)s);
            float time;
            {
                V o, d;
                camera_sample(cam, i, j, s, lp.grid_n, skey, o, d, time);
                stack[0].o = o;
                stack[0].d = d;
                stack[0].thr = mk(1.f, 1.f, 1.f);
                stack[0].key = child_key(skey, BRANCH_ROOT);
                stack[0].depth_flags = 0;
            }
            ++c.primary;
            int top = 0;
            V sample = mk(0, 0, 0);
            while (top >= 0) {
                const PathItem it = stack[top--];
                const int depth = it.depth_flags & 0xffff;
                const bool culling = ((it.depth_flags >> 16) & 1) != 0;
                const bool diffuse_bounce = ((it.depth_flags >> 17) & 1) != 0;
                const V o = it.o, d = it.d, thr = it.thr;
                const uint32_t key = it.key;
                if (!pt && (float)depth >= S.max_depth) {   // Whitted depth limit returns the background (:649-679)
                    sample = add(sample, mul(thr, background(S, cam, d, true, pixelu, pixelv, c)));
                    continue;
                }
                const Hit h = traverse<false, COUNT>(S, o, d, time, culling, S.ieps, 1e30f, c);
                if (h.prim == -1) {
                    // A diffuse-bounce miss under an environment map always fetches one fixed texel: the reference
                    // reads scalar copies of the sampled direction that are still zero (src/ray_trace.cpp:10-12,1093-1116),
                    // which maps to (u, v) = (1, 1/2) lat-long or (1/2, 1/2) probe.
                    if (diffuse_bounce && S.bg_tex >= 0 && S.textures[S.bg_tex].image >= 0) {
                        V bgc = mk(0, 0, 0);
                        tex_sample(S, S.textures[S.bg_tex], S.is_probe ? 0.5f : 1.0f, 0.5f, mk(0, 0, 0), bgc, c);
                        sample = add(sample, mul(thr, bgc));
                    } else {
                        sample = add(sample, mul(thr, background(S, cam, d, !pt, pixelu, pixelv, c)));
                    }
                    continue;
                }
                ++c.shaded;
                const Surface sf = surface_of(S, h);
                const rt_material &m = S.materials[sf.material];
                V p = add(o, scl(d, h.t));
                V normal = normal_of(S, sf, p);
                const float helper_x = sf.kind == 2 ? S.planes[sf.index].raw_normal_x : normal.x;
                apply_normal_map(S, sf, p, normal, c);
                apply_bump_map(S, sf, p, normal, c);
                const float seps = S.seps;
                if (!pt) {
                    // ---- trace_rec -----------------------------------------------------------------
                    if (m.type == RT_MAT_MIRROR || m.type == RT_MAT_CONDUCTOR) {
                        V R = reflect_dir(normal, d, m.roughness, key);
                        V coef = mk(m.mirror);
                        if (m.type == RT_MAT_CONDUCTOR) {
                            float Fs = conductor_fresnel(absf(dot(d, normal)), m.refraction_index, m.absorption_index);
                            coef = mul(mk(Fs, Fs, Fs), mk(m.mirror));
                        }
                        p = add(p, scl(normal, seps));
                        V own = calculate_color<COUNT>(S, normal, m, sf, p, o, time, is_hdr, true, key, c);
                        sample = add(sample, mul(thr, own));
                        ++c.secondary;
                        PathItem &n = stack[++top];
                        n.o = p;
                        n.d = R;
                        n.thr = mul(thr, coef);
                        n.key = child_key(key, BRANCH_REFLECT);
                        n.depth_flags = (depth + 1) | ((int)culling << 16);
                    } else if (m.type == RT_MAT_DIELECTRIC) {
                        float n1 = 1.0f, n2 = m.refraction_index;
                        float cos_theta = dot(d, normal);
                        const bool entering = cos_theta < 0.0f;
                        if (!entering) {
                            float tmp = n1;
                            n1 = n2;
                            n2 = tmp;
                            normal = scl(normal, -1.0f);
                            cos_theta = -cos_theta;
                        }
                        float cos_i = -cos_theta;
                        cos_i = cos_i < 0.0f ? 0.0f : (cos_i > 1.0f ? 1.0f : cos_i);
                        float r0 = rt_div(n1 - n2, n1 + n2);
                        r0 = r0 * r0;
                        const float F = r0 + (1.0f - r0) * rt_pow(1.0f - cos_i, 5.0f);
                        V rdir = reflect_dir(normal, d, m.roughness, key);
                        V tdir;
                        const bool refracts = refract_dir(normal, d, n1, n2, m.roughness, key, tdir);
                        V own = calculate_color<COUNT>(S, normal, m, sf, p, o, time, is_hdr, true, key, c);
                        sample = add(sample, mul(thr, own));
                        if (refracts) {   // pushed first so that the reflected child is walked first, as in the reference
                            V att = mk(1.f, 1.f, 1.f);
                            if (!entering) {
                                V a = scl(mk(m.absorption_coefficient), -h.t);
                                att = mk(rt_exp(a.x), rt_exp(a.y), rt_exp(a.z));
                            }
                            ++c.secondary;
                            PathItem &n = stack[++top];
                            n.o = add(p, scl(normal, -seps));
                            n.d = tdir;
                            n.thr = scl(mul(thr, att), 1.0f - F);
                            n.key = child_key(key, BRANCH_REFRACT);
                            n.depth_flags = (depth + 1);   // refracted rays never cull (:845)
                        }
                        ++c.secondary;
                        PathItem &n = stack[++top];
                        n.o = add(p, scl(normal, seps));
                        n.d = rdir;
                        n.thr = scl(mul(thr, mk(m.mirror)), F);
                        n.key = child_key(key, BRANCH_REFLECT);
                        n.depth_flags = (depth + 1) | ((int)culling << 16);
                    } else {
                        V own = calculate_color<COUNT>(S, normal, m, sf, p, o, time, is_hdr, true, key, c);
                        sample = add(sample, mul(thr, own));
                    }
                    continue;
                }
                // ---- path_trace ---------------------------------------------------------------------
                V kd, ks, kd_rr, rep;
                bool replace_all;
                resolve_textures(S, sf, p, m, kd, ks, kd_rr, replace_all, rep, c);
                if (replace_all) {
                    sample = add(sample, mul(thr, rep));
                    continue;
                }
                V emission = sf.kind == 0 ? mk(S.instances[sf.instance].emission) : (sf.kind == 1 ? mk(S.spheres[sf.index].emission) : mk(0, 0, 0));
                if (m.type != RT_MAT_REGULAR) {
                    if (!((float)depth < S.max_depth)) {
                        sample = add(sample, mul(thr, emission));   // specular surface at the depth limit keeps only its emission
                        continue;
                    }
                    if (m.type == RT_MAT_MIRROR || m.type == RT_MAT_CONDUCTOR) {
                        V R = reflect_dir(normal, d, m.roughness, key);
                        V coef = mk(m.mirror);
                        if (m.type == RT_MAT_CONDUCTOR) {
                            float Fs = conductor_fresnel(absf(dot(d, normal)), m.refraction_index, m.absorption_index);
                            coef = mul(mk(Fs, Fs, Fs), mk(m.mirror));
                        }
                        ++c.secondary;
                        PathItem &n = stack[++top];
                        n.o = add(p, scl(normal, seps));
                        n.d = R;
                        n.thr = mul(thr, coef);
                        n.key = child_key(key, BRANCH_REFLECT);
                        n.depth_flags = (depth + 1) | ((int)culling << 16);
                    } else {
                        float n1 = 1.0f, n2 = m.refraction_index;
                        float cos_theta = dot(d, normal);
                        const bool entering = cos_theta < 0.0f;
                        V rn = normal;
                        if (!entering) {
                            float tmp = n1;
                            n1 = n2;
                            n2 = tmp;
                            rn = scl(normal, -1.0f);
                        } else {
                            cos_theta = -cos_theta;
                        }
                        float r0 = rt_div(n1 - n2, n1 + n2);
                        r0 = r0 * r0;
                        const float F = r0 + (1.0f - r0) * rt_pow(1.0f - cos_theta, 5.0f);
                        V R = reflect_dir(rn, d, m.roughness, key);
                        V T;
                        const bool refracts = refract_dir(rn, d, n1, n2, m.roughness, key, T);
                        if (refracts) {
                            V att = mk(1.f, 1.f, 1.f);
                            if (!entering) {
                                V a = scl(mk(m.absorption_coefficient), -h.t);
                                att = mk(rt_exp(a.x), rt_exp(a.y), rt_exp(a.z));
                            }
                            ++c.secondary;
                            PathItem &n = stack[++top];
                            n.o = add(p, scl(rn, -seps));
                            n.d = T;
                            n.thr = scl(mul(thr, att), 1.0f - F);
                            n.key = child_key(key, BRANCH_REFRACT);
                            n.depth_flags = (depth + 1) | ((int)culling << 16);
                        }
                        ++c.secondary;
                        PathItem &n = stack[++top];
                        n.o = add(p, scl(rn, seps));
                        n.d = R;
                        n.thr = refracts ? scl(mul(thr, mk(m.mirror)), F) : mul(thr, mk(m.mirror));
                        n.key = child_key(key, BRANCH_REFLECT);
                        n.depth_flags = (depth + 1) | ((int)culling << 16);
                    }
                    continue;
                }
                sample = add(sample, mul(thr, emission));
                float rr_prob = 1.0f;
                if (cam.russian_roulette && depth >= cam.min_recursion_depth) {
                    float max_refl = max_std(max_std(kd_rr.x, kd_rr.y), kd_rr.z);
                    rr_prob = min_std(max_refl, 0.99f);
                    if (rnd(key, DIM_RR) > rr_prob) continue;
                }
                if (cam.next_event_estimation) {
                    V direct = calculate_color<COUNT>(S, normal, m, sf, p, o, time, false, false, key, c);
                    sample = add(sample, mul(thr, direct));
                }
                if ((float)depth < S.max_depth) {
                    const bool cosine = cam.importance_sampling != 0;
                    const int split = (depth == 0 && cam.splitting_factor > 1) ? cam.splitting_factor : 1;
                    const V no = add(p, scl(normal, seps));
                    for (int b = split - 1; b >= 0; --b) {   // reverse push: branch 0 is walked first
                        V wi;
                        float pdf;
                        if (split > 1)
                            sample_hemisphere(cosine, rnd(key, DIM_SPLIT + 2 * b), rnd(key, DIM_SPLIT + 2 * b + 1), normal, helper_x, wi, pdf);
                        else
                            sample_hemisphere(cosine, rnd(key, DIM_HEMI_1), rnd(key, DIM_HEMI_2), normal, helper_x, wi, pdf);
                        ++c.secondary;   // the reference traces the bounce even when its pdf rejects the contribution
                        float ndotl = max_std(dot(normal, wi), 0.0f);
                        V w = mk(0, 0, 0);
                        if (pdf > 1e-6f) w = scl(scl(kd, 1.0f / 3.14159265359f), rt_div(ndotl, pdf));
                        w = scl(w, rt_div(1.0f, (float)split));
                        if (rr_prob < 1.0f && rr_prob > 0.0f) w = scl(w, rt_div(1.0f, rr_prob));
                        PathItem &n = stack[++top];
                        n.o = no;
                        n.d = wi;
                        n.thr = mul(thr, w);
                        n.key = child_key(key, BRANCH_BOUNCE + b);
                        n.depth_flags = (depth + 1) | ((int)culling << 16) | (1 << 17);
                    }
                }
            }
            if (cam.sample_max_val > 0.0f) {
                sample.x = min_std(sample.x, cam.sample_max_val);
                sample.y = min_std(sample.y, cam.sample_max_val);
                sample.z = min_std(sample.z, cam.sample_max_val);
            }
            acc0 += sample.x;
            acc1 += sample.y;
            acc2 += sample.z;
        }
        const float ns = (float)lp.n_samples;
        float *dst = out + ((size_t)j * cam.resx + i) * 3;
        dst[0] = rt_div(acc0, ns);
        dst[1] = rt_div(acc1, ns);
        dst[2] = rt_div(acc2, ns);
    }
    // wave-level reduction of the ray counters, one atomic per wave and counter
    uint32_t vals[9] = {c.primary, c.shadow, c.secondary, c.nodes, c.tri, c.sph, c.pln, c.shaded, c.texels};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (!COUNT && (k >= 3 && k <= 6)) continue;
        uint32_t v = wave_sum(vals[k]);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(stats + k, (unsigned long long)v);
    }
}

// Primary visibility only: first sample's ray of each pixel, closest hit (time 0).
template <bool COUNT>
__global__ void __launch_bounds__(256) primary_kernel(DevScene S, rt_camera_desc cam, int grid_n, int32_t *__restrict__ shape_out,
                                                       int32_t *__restrict__ id_out, float *__restrict__ t_out,
                                                       unsigned long long *__restrict__ stats) {
    const int bx = blockIdx.x % ((cam.resx + 15) / 16), by = blockIdx.x / ((cam.resx + 15) / 16);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = bx * 16 + (wave & 1) * 8 + (lane & 7);
    const int j = by * 16 + (wave >> 1) * 8 + (lane >> 3);
    Counters c;
    memset(&c, 0, sizeof c);
    if (i < cam.resx && j < cam.resy) {
        const size_t px = (size_t)j * cam.resx + i;
        const uint32_t skey = sample_key(cam.seed, (uint32_t)px, 0);
        V o, d;
        float time;
        camera_sample(cam, i, j, 0, grid_n, skey, o, d, time);
        ++c.primary;
        const Hit h = traverse<false, COUNT>(S, o, d, 0.0f, false, S.ieps, 1e30f, c);
        int shape = -1, id = -1;
        if (h.prim <= -2) {
            shape = S.n_triangles + S.n_spheres + (-2 - h.prim);
            id = -1;
        } else if (h.prim >= 0) {
            const float4 p2 = S.prims[3 * h.prim + 2];
            shape = __float_as_int(p2.z);
            id = __float_as_int(p2.w);
        }
        shape_out[px] = shape;
        id_out[px] = id;
        t_out[px] = h.prim == -1 ? 1e30f : h.t;
    }
    uint32_t vals[9] = {c.primary, c.shadow, c.secondary, c.nodes, c.tri, c.sph, c.pln, c.shaded, c.texels};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        uint32_t v = wave_sum(vals[k]);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(stats + k, (unsigned long long)v);
    }
}

// Evaluates one of the core's arithmetic definitions per element (self-test hook).
__global__ void math_probe_kernel(int fn, int n, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], r = 0.f;
    switch (fn) {
        case 0: r = rt_sin(x); break;
        case 1: r = rt_cos(x); break;
        case 2: r = rt_asin(x); break;
        case 3: r = rt_acos(x); break;
        case 4: r = rt_atan2(x, y); break;
        case 5: r = rt_exp(x); break;
        case 6: r = rt_log(x); break;
        case 7: r = rt_pow(x, y); break;
        case 8: r = rt_div(x, y); break;
        case 9: r = rt_sqrt(x); break;
        case 10: r = rnd(__float_as_uint(x), __float_as_uint(y)); break;
        case 11: r = x * y + x; break;   // must stay an unfused multiply then add
        default: break;
    }
    out[i] = r;
}

template <class T> int upload(const T *src, size_t n, T **dst) {
    *dst = nullptr;
    if (n == 0) n = 1;   // keep pointers valid for empty arrays
    HIP_OK(hipMalloc((void **)dst, n * sizeof(T)));
    if (src) HIP_OK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

struct rt_scene {
    DevScene dev{};
    std::vector<void *> allocations;
    unsigned long long *stats = nullptr;   // 9 device counters
    float *framebuffer = nullptr;
    size_t framebuffer_floats = 0;
    bool counters = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }

int rt_scene_create(const rt_scene_desc *d, rt_scene **out) {
    if (!d || !out) {
        g_error = "rt_scene_create: null argument";
        return 1;
    }
    if (d->abi_version != RT_ABI_VERSION) {
        g_error = "rt_scene_create: ABI version mismatch";
        return 2;
    }
    // operand checks: every index the kernels follow must be in range
    for (int i = 0; i < d->n_bvh_prims; ++i) {
        const rt_bvh_prim &p = d->bvh_prims[i];
        bool ok = p.shape >= 0 && p.shape < d->n_triangles + d->n_spheres && (p.shape >= d->n_triangles || (p.instance >= 0 && p.instance < d->n_instances));
        if (!ok) {
            g_error = "rt_scene_create: BVH primitive out of range";
            return 3;
        }
    }
    for (int i = 0; i < d->n_bvh_nodes; ++i) {
        const rt_bvh_node &n = d->bvh_nodes[i];
        bool ok = n.n_prims > 0 ? (n.offset >= 0 && n.offset + n.n_prims <= d->n_bvh_prims && n.n_prims <= 0xffff)
                                : (n.offset > i && n.offset < d->n_bvh_nodes && i + 1 < d->n_bvh_nodes && n.axis >= 0 && n.axis < 3);
        if (!ok) {
            g_error = "rt_scene_create: malformed BVH node";
            return 3;
        }
    }
    auto tex_range_ok = [&](int off, int cnt) { return cnt == 0 || (off >= 0 && cnt > 0 && off + cnt <= d->n_texture_refs); };
    for (int i = 0; i < d->n_instances; ++i)
        if (d->instances[i].material < 0 || d->instances[i].material >= d->n_materials || !tex_range_ok(d->instances[i].tex_offset, d->instances[i].tex_count)) {
            g_error = "rt_scene_create: instance material/texture index out of range";
            return 3;
        }
    for (int i = 0; i < d->n_spheres; ++i)
        if (d->spheres[i].material < 0 || d->spheres[i].material >= d->n_materials || !tex_range_ok(d->spheres[i].tex_offset, d->spheres[i].tex_count)) {
            g_error = "rt_scene_create: sphere material/texture index out of range";
            return 3;
        }
    for (int i = 0; i < d->n_planes; ++i)
        if (d->planes[i].material < 0 || d->planes[i].material >= d->n_materials || !tex_range_ok(d->planes[i].tex_offset, d->planes[i].tex_count)) {
            g_error = "rt_scene_create: plane material/texture index out of range";
            return 3;
        }
    for (int i = 0; i < d->n_texture_refs; ++i)
        if (d->texture_refs[i] < 0 || d->texture_refs[i] >= d->n_textures) {
            g_error = "rt_scene_create: texture reference out of range";
            return 3;
        }
    for (int i = 0; i < d->n_textures; ++i)
        if (d->textures[i].image >= d->n_images) {
            g_error = "rt_scene_create: texture image out of range";
            return 3;
        }
    for (int i = 0; i < d->n_images; ++i) {
        const rt_image &im = d->images[i];
        if (im.width <= 0 || im.height <= 0) continue;
        int64_t need = (int64_t)im.offset + (int64_t)im.width * im.height * (im.is_hdr ? 4 : 3);
        if (im.offset < 0 || need > (im.is_hdr ? d->n_texels_f32 : d->n_texels_u8)) {
            g_error = "rt_scene_create: image texels out of range";
            return 3;
        }
    }
    for (int i = 0; i < d->n_lights; ++i) {
        const rt_light &l = d->lights[i];
        if (l.samples < 1 || (l.type == RT_LIGHT_SPHERE && (l.sphere < 0 || l.sphere >= d->n_spheres)) ||
            (l.type == RT_LIGHT_ENV && l.image >= d->n_images)) {
            g_error = "rt_scene_create: malformed light";
            return 3;
        }
    }
    if (d->background_texture >= d->n_textures) {
        g_error = "rt_scene_create: background texture out of range";
        return 3;
    }

    rt_scene *s = new rt_scene();
    HIP_OK(hipGetDevice(&s->device));
    DevScene &D = s->dev;
    auto keep = [&](void *p) { s->allocations.push_back(p); };

    // BVH nodes -> 2 x float4
    std::vector<float4> nodes((size_t)d->n_bvh_nodes * 2);
    for (int i = 0; i < d->n_bvh_nodes; ++i) {
        const rt_bvh_node &n = d->bvh_nodes[i];
        int meta = n.n_prims > 0 ? n.n_prims : (n.axis << 16);
        float fo, fm;
        memcpy(&fo, &n.offset, 4);
        memcpy(&fm, &meta, 4);
        nodes[2 * i] = make_float4(n.bmin[0], n.bmin[1], n.bmin[2], fo);
        nodes[2 * i + 1] = make_float4(n.bmax[0], n.bmax[1], n.bmax[2], fm);
    }
    // leaf entries -> 3 x float4 (scaled vertices and edges as the reference derives them, src/shapes.cpp:22-29,236-237)
    std::vector<float4> prims((size_t)d->n_bvh_prims * 3);
    auto as_float = [](int v) {
        float f;
        memcpy(&f, &v, 4);
        return f;
    };
    for (int i = 0; i < d->n_bvh_prims; ++i) {
        const rt_bvh_prim &p = d->bvh_prims[i];
        if (p.shape < d->n_triangles) {
            const rt_triangle &t = d->triangles[p.shape];
            float c1s[3], e1[3], e2[3];
            for (int a = 0; a < 3; ++a) {
                c1s[a] = t.c1[a] * 1000.0f;
                e1[a] = t.c2[a] * 1000.0f - c1s[a];
                e2[a] = t.c3[a] * 1000.0f - c1s[a];
            }
            prims[3 * i] = make_float4(c1s[0], c1s[1], c1s[2], e1[0]);
            prims[3 * i + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
            prims[3 * i + 2] = make_float4(e2[2], as_float(p.instance), as_float(p.shape), as_float(d->instances[p.instance].ref_id));
        } else {
            prims[3 * i] = make_float4(0, 0, 0, 0);
            prims[3 * i + 1] = make_float4(0, 0, 0, 0);
            prims[3 * i + 2] = make_float4(0, as_float(-1 - (p.shape - d->n_triangles)), as_float(p.shape), as_float(123456));
        }
    }
    std::vector<float4> inst((size_t)d->n_instances * 4);
    for (int i = 0; i < d->n_instances; ++i) {
        const rt_instance &in = d->instances[i];
        int flags = (in.identity ? 1 : 0) | ((in.motionblur[0] != 0.f || in.motionblur[1] != 0.f || in.motionblur[2] != 0.f) ? 2 : 0);
        for (int r = 0; r < 3; ++r) inst[4 * i + r] = make_float4(in.inv_model[4 * r], in.inv_model[4 * r + 1], in.inv_model[4 * r + 2], in.inv_model[4 * r + 3]);
        inst[4 * i + 3] = make_float4(in.motionblur[0], in.motionblur[1], in.motionblur[2], as_float(flags));
    }
    // flat normals: normalize(cross(e1s, e2s)) (src/shapes.cpp:26-33), evaluated in the same order as on the device
    std::vector<float> flat((size_t)d->n_triangles * 3);
    for (int i = 0; i < d->n_triangles; ++i) {
        const rt_triangle &t = d->triangles[i];
        float e1[3], e2[3];
        for (int a = 0; a < 3; ++a) {
            float c1s = t.c1[a] * 1000.0f;
            e1[a] = t.c2[a] * 1000.0f - c1s;
            e2[a] = t.c3[a] * 1000.0f - c1s;
        }
        float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        float len2 = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
        float inv = len2 > 1e-12f ? 1.0f / sqrtf(len2) : 0.0f;
        for (int a = 0; a < 3; ++a) flat[3 * i + a] = len2 > 1e-12f ? n[a] * inv : 0.0f;
    }

#define UP(field, vec_ptr, count)                                                     \
    do {                                                                              \
        decltype(D.field) dp_ = nullptr;                                              \
        typedef typename std::remove_const<typename std::remove_pointer<decltype(D.field)>::type>::type T_; \
        T_ *tmp_ = nullptr;                                                           \
        if (upload<T_>((const T_ *)(vec_ptr), (size_t)(count), &tmp_)) {              \
            rt_scene_destroy(s);                                                      \
            return 10;                                                                \
        }                                                                             \
        dp_ = tmp_;                                                                   \
        D.field = dp_;                                                                \
        keep(tmp_);                                                                   \
    } while (0)
    UP(nodes, nodes.data(), nodes.size());
    UP(prims, prims.data(), prims.size());
    UP(inst, inst.data(), inst.size());
    UP(instances, d->instances, d->n_instances);
    UP(triangles, d->triangles, d->n_triangles);
    UP(flat_normals, flat.data(), flat.size());
    UP(spheres, d->spheres, d->n_spheres);
    UP(planes, d->planes, d->n_planes);
    UP(materials, d->materials, d->n_materials);
    UP(textures, d->textures, d->n_textures);
    UP(texture_refs, d->texture_refs, d->n_texture_refs);
    UP(images, d->images, d->n_images);
    UP(texels_u8, d->texels_u8, d->n_texels_u8);
    UP(texels_f32, d->texels_f32, d->n_texels_f32);
    UP(lights, d->lights, d->n_lights);
#undef UP
    D.n_triangles = d->n_triangles;
    D.n_spheres = d->n_spheres;
    D.n_planes = d->n_planes;
    D.n_lights = d->n_lights;
    D.n_nodes = d->n_bvh_nodes;
    D.ieps = d->intersection_epsilon;
    D.seps = d->shadow_ray_epsilon;
    D.max_depth = d->max_depth;
    memcpy(D.ambient, d->ambient_light, sizeof D.ambient);
    memcpy(D.bg, d->background_color, sizeof D.bg);
    D.bg_tex = d->background_texture;
    D.is_probe = d->is_probe;
    if (hipMalloc((void **)&s->stats, 9 * sizeof(unsigned long long)) != hipSuccess || hipStreamCreate(&s->stream) != hipSuccess ||
        hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) {
        g_error = "rt_scene_create: cannot allocate stream/events";
        rt_scene_destroy(s);
        return 10;
    }
    *out = s;
    return 0;
}

void rt_scene_destroy(rt_scene *s) {
    if (!s) return;
    for (void *p : s->allocations) (void)hipFree(p);
    if (s->stats) (void)hipFree(s->stats);
    if (s->framebuffer) (void)hipFree(s->framebuffer);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int rt_set_counters(rt_scene *s, int32_t enabled) {
    if (!s) return 1;
    s->counters = enabled != 0;
    return 0;
}

static int check_camera(const rt_scene *s, const rt_camera_desc *cam, int &grid_n, int &n_samples) {
    if (cam->resx < 1 || cam->resy < 1 || cam->num_samples < 1) {
        g_error = "invalid camera: resolution and sample count must be positive";
        return 4;
    }
    grid_n = 1;
    n_samples = 1;
    if (cam->num_samples != 1) {
        grid_n = (int)sqrt((double)cam->num_samples);   // src/camera.cpp:149
        n_samples = grid_n * grid_n;
    }
    int split = cam->splitting_factor > 1 ? cam->splitting_factor : 1;
    // pending-ray stack bound: at most one extra item per depth level, plus the depth-0 split fan-out
    if ((int)s->dev.max_depth + 2 + split > PATH_STACK) {
        g_error = "MaxRecursionDepth + SplittingFactor exceeds the per-lane path stack (" + std::to_string(PATH_STACK) + ")";
        return 5;
    }
    if (s->dev.max_depth > 60000.f) {
        g_error = "MaxRecursionDepth too large";
        return 5;
    }
    return 0;
}

static void read_stats(rt_scene *s, rt_stats *st, float kernel_ms, float total_ms) {
    if (!st) return;
    unsigned long long v[9];
    (void)hipMemcpy(v, s->stats, sizeof v, hipMemcpyDeviceToHost);
    memset(st, 0, sizeof *st);
    st->primary_rays = v[0];
    st->shadow_rays = v[1];
    st->secondary_rays = v[2];
    st->nodes_visited = v[3];
    st->triangle_tests = v[4];
    st->sphere_tests = v[5];
    st->plane_tests = v[6];
    st->shaded_hits = v[7];
    st->texel_fetches = v[8];
    st->kernel_ms = kernel_ms;
    st->total_ms = total_ms;
}

int rt_render_shard(rt_scene *s, const rt_camera_desc *cam, int32_t rank, int32_t world, float *hdr_out, rt_stats *stats) {
    if (!s || !cam || !hdr_out) {
        g_error = "rt_render: null argument";
        return 1;
    }
    if (world < 1 || rank < 0 || rank >= world) {
        g_error = "rt_render_shard: rank/world out of range";
        return 1;
    }
    int grid_n, n_samples;
    if (int rc = check_camera(s, cam, grid_n, n_samples)) return rc;
    auto t0 = std::chrono::high_resolution_clock::now();
    HIP_OK(hipSetDevice(s->device));
    LaunchParams lp;
    lp.tiles_x = (cam->resx + TILE - 1) / TILE;
    lp.tiles_y = (cam->resy + TILE - 1) / TILE;
    lp.rank = rank;
    lp.world = world;
    const int total_tiles = lp.tiles_x * lp.tiles_y;
    lp.n_owned = total_tiles > rank ? (total_tiles - rank + world - 1) / world : 0;
    lp.grid_n = grid_n;
    lp.n_samples = n_samples;

    // output: device pointer -> write in place; host pointer -> internal framebuffer + copy back
    hipPointerAttribute_t attr;
    bool out_on_device = false;
    if (hipPointerGetAttributes(&attr, hdr_out) == hipSuccess) out_on_device = attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    const size_t n_floats = (size_t)cam->resx * cam->resy * 3;
    float *dev_out = hdr_out;
    if (!out_on_device) {
        if (s->framebuffer_floats < n_floats) {
            if (s->framebuffer) (void)hipFree(s->framebuffer);
            s->framebuffer = nullptr;
            s->framebuffer_floats = 0;
            HIP_OK(hipMalloc((void **)&s->framebuffer, n_floats * sizeof(float)));
            s->framebuffer_floats = n_floats;
        }
        dev_out = s->framebuffer;
        if (world > 1) HIP_OK(hipMemcpyAsync(dev_out, hdr_out, n_floats * sizeof(float), hipMemcpyHostToDevice, s->stream));
    }
    HIP_OK(hipMemsetAsync(s->stats, 0, 9 * sizeof(unsigned long long), s->stream));
    HIP_OK(hipEventRecord(s->ev0, s->stream));
    if (lp.n_owned > 0) {
        dim3 grid((unsigned)lp.n_owned * 4), block(BLOCK_EDGE * BLOCK_EDGE);
        if (s->counters)
            hipLaunchKernelGGL(render_kernel<true>, grid, block, 0, s->stream, s->dev, *cam, lp, dev_out, s->stats);
        else
            hipLaunchKernelGGL(render_kernel<false>, grid, block, 0, s->stream, s->dev, *cam, lp, dev_out, s->stats);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(s->ev1, s->stream));
    if (!out_on_device) HIP_OK(hipMemcpyAsync(hdr_out, dev_out, n_floats * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    float kernel_ms = 0.f;
    HIP_OK(hipEventElapsedTime(&kernel_ms, s->ev0, s->ev1));
    auto t1 = std::chrono::high_resolution_clock::now();
    read_stats(s, stats, kernel_ms, (float)std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}

int rt_render(rt_scene *s, const rt_camera_desc *cam, float *hdr_out, rt_stats *stats) { return rt_render_shard(s, cam, 0, 1, hdr_out, stats); }

int rt_trace_primary(rt_scene *s, const rt_camera_desc *cam, int32_t *shape_out, int32_t *id_out, float *t_out, rt_stats *stats) {
    if (!s || !cam || !shape_out || !id_out || !t_out) {
        g_error = "rt_trace_primary: null argument";
        return 1;
    }
    int grid_n, n_samples;
    if (int rc = check_camera(s, cam, grid_n, n_samples)) return rc;
    HIP_OK(hipSetDevice(s->device));
    const size_t n = (size_t)cam->resx * cam->resy;
    int32_t *d_shape = nullptr, *d_id = nullptr;
    float *d_t = nullptr;
    HIP_OK(hipMalloc((void **)&d_shape, n * 4));
    HIP_OK(hipMalloc((void **)&d_id, n * 4));
    HIP_OK(hipMalloc((void **)&d_t, n * 4));
    HIP_OK(hipMemsetAsync(s->stats, 0, 9 * sizeof(unsigned long long), s->stream));
    dim3 grid((unsigned)(((cam->resx + 15) / 16) * ((cam->resy + 15) / 16))), block(256);
    HIP_OK(hipEventRecord(s->ev0, s->stream));
    if (s->counters)
        hipLaunchKernelGGL(primary_kernel<true>, grid, block, 0, s->stream, s->dev, *cam, grid_n, d_shape, d_id, d_t, s->stats);
    else
        hipLaunchKernelGGL(primary_kernel<false>, grid, block, 0, s->stream, s->dev, *cam, grid_n, d_shape, d_id, d_t, s->stats);
    hipError_t le = hipGetLastError();
    HIP_OK(hipEventRecord(s->ev1, s->stream));
    hipError_t e1 = hipMemcpyAsync(shape_out, d_shape, n * 4, hipMemcpyDeviceToHost, s->stream);
    hipError_t e2 = hipMemcpyAsync(id_out, d_id, n * 4, hipMemcpyDeviceToHost, s->stream);
    hipError_t e3 = hipMemcpyAsync(t_out, d_t, n * 4, hipMemcpyDeviceToHost, s->stream);
    hipError_t e4 = hipStreamSynchronize(s->stream);
    (void)hipFree(d_shape);
    (void)hipFree(d_id);
    (void)hipFree(d_t);
    HIP_OK(le);
    HIP_OK(e1);
    HIP_OK(e2);
    HIP_OK(e3);
    HIP_OK(e4);
    float kernel_ms = 0.f;
    HIP_OK(hipEventElapsedTime(&kernel_ms, s->ev0, s->ev1));
    read_stats(s, stats, kernel_ms, kernel_ms);
    return 0;
}

int rt_selftest_math(int32_t fn, int32_t n, const float *a, const float *b, float *out) {
    if (n <= 0 || !a || !b || !out) {
        g_error = "rt_selftest_math: bad argument";
        return 1;
    }
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_OK(hipMalloc((void **)&da, (size_t)n * 4));
    HIP_OK(hipMalloc((void **)&db, (size_t)n * 4));
    HIP_OK(hipMalloc((void **)&dout, (size_t)n * 4));
    HIP_OK(hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, n, da, db, dout);
    hipError_t le = hipGetLastError();
    hipError_t ce = hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    HIP_OK(le);
    HIP_OK(ce);
    return 0;
}

}  // extern "C"
