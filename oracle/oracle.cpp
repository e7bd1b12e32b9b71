// TEST INFRASTRUCTURE — CPU oracle.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product never does.
//
// A scalar, strict-IEEE restatement of the reference renderer's hot path over the flat scene
// arrays of include/gpu_render.h.  Every function cites the reference file:line it follows.
// Pinned against the compiled reference (oracle/_ref/ref_dump) by tests/test_oracle_vs_reference.py
// through the fixtures in tests/golden/.
//
// Deliberate, documented departures from the reference (none changes the algorithm):
//  * vec3 normalisation is x * (1/sqrt(len2)) instead of _mm_rsqrt_ps + one Newton step
//    (src/algebra.cpp:50-65); the zero-vector guard len2 > 1e-12 is kept.
//  * libm / sse_mathfun calls are replaced by the deterministic functions of oracle_math.h.
//  * get_random_float() (unseedable thread-local mt19937, src/helper.cpp:21-26) is replaced by a
//    counter-based generator keyed by (seed, pixel, sample, path node, dimension).
//  * the thread-local last-shadow-blocker cache (src/ray_trace.cpp:158-159,207-226) is not modelled;
//    it is an optimisation that changes results only for a first-found blocker inside the last
//    shadow-ray epsilon.
//  * vec3's stale scalar shadow copies (src/algebra.h:8-62) are modelled only where they change
//    control flow (Russian-roulette reflectance under blend_kd, ONB helper-axis choice after normal
//    mapping); Perlin textures always see the current hit point.
//  * instance inverse matrices come with the scene description.  The host ingest makes them as the reference's
//    Cramer inverse does (src/algebra.cpp:246-370: cofactors times the approximate reciprocal of the determinant refined by
//    one Newton step, the instruction's estimate modelled as 1 / d cut to 12 bits), so the inverse of the identity is
//    (1 - 2^-24) I here as there, and the triangle test applies it (src/shapes.cpp:224-230).
//
// Besides the reference's walk (Tracer::intersect) the library can walk the PRODUCT's wide tree on the host
// (Tracer::intersect_wide over the records of raytracer_amd/csrc/gpu/rt_wide.h, the one product header it includes):
// orc_wide_check validates the records against the two-child tree, orc_intersect_many / orc_set_wide(2) compare the
// two walks ray by ray.  That checks the product's data structure with the checker's own primitive tests; the product
// does not use any of it.
// Compile with -ffp-contract=off and without fast-math.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "gpu_render.h"
#include "oracle_math.h"
#include "rt_wide.h"   // the product's wide-tree builder: the checker walks the records the product walks

namespace {

// ---------------------------------------------------------------------------------------------
// vec3 arithmetic in the reference's SSE evaluation order (src/algebra.cpp:5-75)
// ---------------------------------------------------------------------------------------------
struct V {
    float x, y, z;
};
inline V mk(float a, float b, float c) { return V{a, b, c}; }
inline V mk(const float *p) { return V{p[0], p[1], p[2]}; }
inline V add(V a, V b) { return V{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V sub(V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V mul(V a, V b) { return V{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V scl(V a, float s) { return V{a.x * s, a.y * s, a.z * s}; }
// _mm_dp_ps: (x*x' + y*y') + (z*z' + 0)
inline float dot(V a, V b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V cross(V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V normalize(V a) {   // src/algebra.cpp:50-65 (exact reciprocal square root, see header)
    float d = dot(a, a);
    if (!(d > 1e-12f)) return V{0.f, 0.f, 0.f};
    float inv = 1.0f / sqrtf(d);
    return scl(a, inv);
}
inline float fmax_(float a, float b) { return a < b ? b : a; }   // std::max(a, b)
inline float fmin_(float a, float b) { return b < a ? b : a; }   // std::min(a, b)
// simd_mat4::mult_vec (src/algebra.cpp:217-244): (m0*x + m1*y) + (m2*z + m3*w)
inline V xpoint(const float *m, V v) {
    return V{(m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3]), (m[4] * v.x + m[5] * v.y) + (m[6] * v.z + m[7]),
             (m[8] * v.x + m[9] * v.y) + (m[10] * v.z + m[11])};
}
inline V xdir(const float *m, V v) {
    return V{(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z, (m[8] * v.x + m[9] * v.y) + m[10] * v.z};
}

// ---------------------------------------------------------------------------------------------
// counter-based random numbers (replaces src/helper.cpp:21-26)
// ---------------------------------------------------------------------------------------------
inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
inline uint32_t sample_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return hash32(hash32(hash32(seed + 0x9e3779b9u) ^ pixel) ^ (sample * 0x85ebca6bu + 1u));
}
inline uint32_t child_key(uint32_t key, uint32_t branch) { return hash32(key + 0x9e3779b9u * (branch + 1u)); }
inline float rnd(uint32_t key, uint32_t dim) { return (float)(hash32(key ^ hash32(dim + 0x68bc21ebu)) >> 8) * (1.0f / 16777216.0f); }

enum {   // dimensions inside one path node
    DIM_REFLECT_XI1 = 0, DIM_REFLECT_XI2 = 1, DIM_REFRACT_XI1 = 2, DIM_REFRACT_XI2 = 3, DIM_RR = 4, DIM_HEMI_1 = 5, DIM_HEMI_2 = 6,
    DIM_SPLIT = 8, DIM_LIGHT = 4096,
    // dimensions of the camera sample key
    DIM_CAM_R1 = 0, DIM_CAM_R2 = 1, DIM_CAM_R3 = 2, DIM_CAM_R4 = 3, DIM_CAM_TIME = 4,
    BRANCH_ROOT = 0, BRANCH_REFLECT = 1, BRANCH_REFRACT = 2, BRANCH_BOUNCE = 3
};

struct TriAccel {   // derived once per triangle (src/shapes.cpp:19-48, 236-237)
    V c1s, e1, e2, flat_normal;
};

struct Counters {
    uint64_t primary = 0, shadow = 0, secondary = 0, nodes = 0, tri = 0, sph = 0, pln = 0, shaded = 0, texels = 0;
    void add(const Counters &o) {
        primary += o.primary;
        shadow += o.shadow;
        secondary += o.secondary;
        nodes += o.nodes;
        tri += o.tri;
        sph += o.sph;
        pln += o.pln;
        shaded += o.shaded;
        texels += o.texels;
    }
};

}  // namespace

struct orc_scene {
    rt_scene_desc d;
    std::vector<rt_triangle> triangles;
    std::vector<rt_instance> instances;
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<int32_t> texture_refs;
    std::vector<rt_image> images;
    std::vector<uint8_t> texels_u8;
    std::vector<float> texels_f32;
    std::vector<rt_light> lights;
    std::vector<rt_bvh_node> nodes;
    std::vector<rt_bvh_prim> prims;
    std::vector<TriAccel> accel;
    // the product's wide tree (raytracer_amd/csrc/gpu/rt_wide.h), built on demand by orc_intersect_wide
    std::vector<rt::WideRecord> wide;
    std::vector<int> leaf_node_at;   // first leaf entry -> the leaf's node
    std::vector<int> wide_slot_node;   // study: record * 4 + slot -> two-child node (-1 empty), filled by orc_wide_check
    int wide_state = 0;   // 0 not built yet, 1 built, -1 the tree does not qualify
};

namespace {

struct Hit {
    bool hit = false;
    float t = 0.f;
    int shape = -1;      // position in get_shapes() order: triangles, spheres, planes
    int instance = -1;   // index into instances (triangles only)
};

std::atomic<long> g_walk_mismatches{0};
std::vector<int> *g_wide_log = nullptr;   // diagnostic: records (>= 0) and leaves (-1 - first entry) a wide walk touches, -2^30 after each ray
// work of the walks on the calling thread (read by orc_walk_work after a single-threaded render)
thread_local long g_wide_steps = 0, g_wide_leaf_arrivals = 0, g_wide_leaf_visits = 0, g_two_interior = 0, g_wide_pushes = 0, g_wide_maxsp = 0;
int g_walk_wide = 0;
int g_walk_exact_slots = 0;   // study: the wide walk tests the slots' exact boxes instead of the quantised ones (what does 8 bits cost?)   // orc_set_wide: renders walk the product's wide tree instead of the two-child tree

struct Tracer {
    const orc_scene &S;
    const rt_camera_desc &cam;
    Counters cnt;
    Tracer(const orc_scene &s, const rt_camera_desc &c) : S(s), cam(c) {}

    // ---- primitives ----------------------------------------------------------------------
    // triangle::intersect, src/shapes.cpp:212-279
    bool tri_intersect(int tri, int inst_idx, V o, V d, float time, bool culling, float eps, float &t) {
        ++cnt.tri;
        const rt_instance &in = S.instances[inst_idx];
        const TriAccel &a = S.accel[tri];
        V offset = scl(mk(in.motionblur), time);
        V moved = sub(o, offset);
        V oo, dd;
        bool scaled_identity = in.identity != 0;
        for (int r = 0; r < 3 && scaled_identity; ++r)
            for (int c = 0; c < 4; ++c)
                if (in.inv_model[4 * r + c] != (r == c ? in.inv_model[0] : 0.f)) scaled_identity = false;
        if (scaled_identity) {
            // the inverse of the identity as the reference computes it is k I, k = 1 - 2^-24 (src/algebra.cpp:345-347):
            // the matrix product of src/shapes.cpp:227-230 with the zero terms left out
            oo = scl(moved, in.inv_model[0]);
            dd = scl(d, in.inv_model[0]);
        } else {
            oo = xpoint(in.inv_model, moved);
            dd = xdir(in.inv_model, d);
        }
        V os = scl(oo, 1000.0f);
        V h = cross(dd, a.e2);
        float det = dot(a.e1, h);
        if (culling) {
            if (det < eps) return false;
        } else {
            if (fabsf(det) < eps) return false;
        }
        float f = 1.0f / det;
        V s = sub(os, a.c1s);
        float u = dot(s, h);
        u *= f;
        if (u < (0.0f - eps) || u > (1.0f + eps)) return false;
        V q = cross(s, a.e1);
        float v = dot(dd, q);
        v *= f;
        if (v < (0.0f - eps) || (u + v) > (1.0f + eps)) return false;
        t = dot(a.e2, q);
        t *= f;
        t *= (1.0f / 1000.0f);
        return t > eps;
    }
    // sphere::intersect, src/shapes.cpp:407-447
    bool sphere_intersect(int si, V o, V d, float eps, float &t) {
        ++cnt.sph;
        const rt_sphere &s = S.spheres[si];
        V oo = xpoint(s.inv_model, o);
        V dd = xdir(s.inv_model, d);
        V oc = sub(oo, mk(s.center));
        float a = dot(dd, dd);
        float b = dot(oc, dd);
        b *= 2.0f;
        float c = dot(oc, oc);
        c -= (s.radius * s.radius);
        float disc = b * b - 4 * a * c;
        if (disc < 0.0f) return false;
        float sq = sqrtf(disc);
        float inv2a = 1.0f / (2.0f * a);
        float t1 = (-b - sq) * inv2a;
        float t2 = (-b + sq) * inv2a;
        if (t1 > eps)
            t = t1;
        else if (t2 > eps)
            t = t2;
        else
            return false;
        return true;
    }
    // plane::intersect, src/shapes.cpp:506-527
    bool plane_intersect(int pi, V o, V d, float eps, float &t) {
        ++cnt.pln;
        const rt_plane &p = S.planes[pi];
        V n = mk(p.normal);
        float denom = dot(n, d);
        if (fabsf(denom) < eps) return false;
        V p0l0 = sub(mk(p.point), o);
        float num = dot(p0l0, n);
        t = num / denom;
        if (t < eps) return false;
        return true;
    }

    // ---- BVH -----------------------------------------------------------------------------
    // bvh::intersect_aabb_fast, src/bvh.cpp:348-381
    static bool slab(const rt_bvh_node &n, V o, V inv, const int neg[3], float t_hit) {
        float tmin = ((neg[0] ? n.bmax[0] : n.bmin[0]) - o.x) * inv.x;
        float tmax = ((neg[0] ? n.bmin[0] : n.bmax[0]) - o.x) * inv.x;
        float tymin = ((neg[1] ? n.bmax[1] : n.bmin[1]) - o.y) * inv.y;
        float tymax = ((neg[1] ? n.bmin[1] : n.bmax[1]) - o.y) * inv.y;
        if ((tmin > tymax) || (tymin > tmax)) return false;
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        float tzmin = ((neg[2] ? n.bmax[2] : n.bmin[2]) - o.z) * inv.z;
        float tzmax = ((neg[2] ? n.bmin[2] : n.bmax[2]) - o.z) * inv.z;
        if ((tmin > tzmax) || (tzmin > tmax)) return false;
        if (tzmin > tmin) tmin = tzmin;
        if (tzmax < tmax) tmax = tzmax;
        return (tmin < t_hit) && (tmax > 0.0f);
    }
    // bvh::intersect, src/bvh.cpp:383-480
    Hit intersect(V o, V d, float time, bool culling, float eps, bool any_hit, float stop_t) {
        if (g_walk_wide == 1 && S.wide_state == 1) return intersect_wide(o, d, time, culling, eps, any_hit, stop_t);
        if (g_walk_wide == 3 && S.wide_state == 1) {   // logging walk (orc_wide_log)
            const Hit h = intersect_wide(o, d, time, culling, eps, any_hit, stop_t);
            if (g_wide_log) g_wide_log->push_back(any_hit ? -(1 << 30) - 1 : -(1 << 30));
            return h;
        }
        if (g_walk_wide == 2 && S.wide_state == 1) {   // both walks: report rays on which they disagree
            g_walk_wide = 0;
            const Counters keep = cnt;
            const Hit a = intersect(o, d, time, culling, eps, any_hit, stop_t);
            cnt = keep;
            g_walk_wide = 2;
            const Hit b = intersect_wide(o, d, time, culling, eps, any_hit, stop_t);
            if (a.hit != b.hit || a.shape != b.shape || a.instance != b.instance || memcmp(&a.t, &b.t, 4) != 0) {
                if (g_walk_mismatches++ < 8)
                    fprintf(stderr, "walks disagree: o %.9g %.9g %.9g d %.9g %.9g %.9g any %d stop %.9g eps %g: two-child %d/%d t %.9g, wide %d/%d t %.9g\n",
                            o.x, o.y, o.z, d.x, d.y, d.z, (int)any_hit, stop_t, eps, a.hit ? a.shape : -1, a.instance, a.t, b.hit ? b.shape : -1, b.instance, b.t);
            }
            return a;
        }
        Hit h;
        h.t = stop_t;
        const int ntri = (int)S.triangles.size(), nsph = (int)S.spheres.size();
        for (int pi = 0; pi < (int)S.planes.size(); ++pi) {
            float tc;
            if (plane_intersect(pi, o, d, eps, tc)) {
                if (tc > eps && tc < h.t) {
                    h.t = tc;
                    h.hit = true;
                    h.shape = ntri + nsph + pi;
                    h.instance = -1;
                    if (any_hit) return h;
                }
            }
        }
        if (S.nodes.empty()) return h;
        V inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        int neg[3] = {inv.x < 0, inv.y < 0, inv.z < 0};
        int stack[64];
        int top = 0;
        stack[0] = 0;
        while (top >= 0) {
            int ni = stack[top--];
            const rt_bvh_node &n = S.nodes[ni];
            ++cnt.nodes;
            if (!slab(n, o, inv, neg, h.t)) continue;
            if (n.n_prims > 0) {
                for (int i = 0; i < n.n_prims; ++i) {
                    const rt_bvh_prim &pr = S.prims[n.offset + i];
                    float tc;
                    bool ok = pr.shape < ntri ? tri_intersect(pr.shape, pr.instance, o, d, time, culling, eps, tc)
                                              : sphere_intersect(pr.shape - ntri, o, d, eps, tc);
                    if (ok && tc > eps && tc < h.t) {
                        h.t = tc;
                        h.hit = true;
                        h.shape = pr.shape;
                        h.instance = pr.instance;
                        if (any_hit) return h;
                    }
                }
            } else {
                ++g_two_interior;
                int near_i, far_i;
                if (neg[n.axis]) {
                    near_i = n.offset;
                    far_i = ni + 1;
                } else {
                    near_i = ni + 1;
                    far_i = n.offset;
                }
                stack[++top] = far_i;
                stack[++top] = near_i;
            }
        }
        return h;
    }

    // The walk of the product's wide tree (wide_step in raytracer_amd/csrc/gpu/rt_device.h), one ray at a time on the
    // host: same record decoding, same arithmetic, same visit order; leaves are tested as in intersect() above.  Used
    // to check that the wide walk finds what the reference's two-child walk finds.
    Hit intersect_wide(V o, V d, float time, bool culling, float eps, bool any_hit, float stop_t) {
        Hit h;
        h.t = stop_t;
        const int ntri = (int)S.triangles.size(), nsph = (int)S.spheres.size();
        for (int pi = 0; pi < (int)S.planes.size(); ++pi) {
            float tc;
            if (plane_intersect(pi, o, d, eps, tc)) {
                if (tc > eps && tc < h.t) {
                    h.t = tc;
                    h.hit = true;
                    h.shape = ntri + nsph + pi;
                    h.instance = -1;
                    if (any_hit) return h;
                }
            }
        }
        if (S.nodes.empty()) return h;
        V inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        int neg[3] = {inv.x < 0, inv.y < 0, inv.z < 0};
        ++cnt.nodes;
        if (!slab(S.nodes[0], o, inv, neg, h.t)) return h;
        const unsigned sgn = (unsigned)(neg[0] | neg[1] << 1 | neg[2] << 2) * 0x49u;
        struct Entry {
            int ref;
            float t;
        };
        Entry stack[64];
        int top = 0;
        int cur = 0;
        const int POP = 0x40000000;
        const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
        for (;;) {
            if (cur == POP) {
                bool got = false;
                while (top > 0) {
                    --top;
                    if (any_hit || stack[top].t < h.t) {
                        cur = stack[top].ref;
                        got = true;
                        break;
                    }
                }
                if (!got) return h;
            }
            if (cur < 0) {   // leaf reference: 1 << 31 | first entry (on the device the last entry carries an end mark)
                const int first = cur & 0x7fffffff, n = S.nodes[(size_t)S.leaf_node_at[(size_t)first]].n_prims;
                // the leaf's own (exact) box decides whether its entries are tested, as in the two-child walk: the
                // quantised boxes only decide what is looked at
                ++cnt.nodes;
                ++g_wide_leaf_arrivals;
                if (g_wide_log) g_wide_log->push_back(-1 - first);
                if (!slab(S.nodes[(size_t)S.leaf_node_at[(size_t)first]], o, inv, neg, h.t)) {
                    cur = POP;
                    continue;
                }
                ++g_wide_leaf_visits;
                for (int i = 0; i < n; ++i) {
                    const rt_bvh_prim &pr = S.prims[first + i];
                    float tc;
                    bool ok = pr.shape < ntri ? tri_intersect(pr.shape, pr.instance, o, d, time, culling, eps, tc)
                                              : sphere_intersect(pr.shape - ntri, o, d, eps, tc);
                    if (ok && tc > eps && tc < h.t) {
                        h.t = tc;
                        h.hit = true;
                        h.shape = pr.shape;
                        h.instance = pr.instance;
                        if (any_hit) return h;
                    }
                }
                cur = POP;
                continue;
            }
            ++g_wide_steps;
            if (g_wide_log) g_wide_log->push_back(cur);
            const float *f = S.wide[(size_t)cur].f;
            unsigned w[16];
            memcpy(w, f, sizeof w);
            const unsigned head = w[3];
            float scale[3];
            const unsigned sx_bits = head & 0x7f800000u;
            memcpy(&scale[0], &sx_bits, 4);
            scale[1] = f[14], scale[2] = f[15];
            const unsigned along = sgn & head;
            const bool s_root = (along & 7u) != 0, s_l = (along & 0x38u) != 0, s_r = (along & 0x1c0u) != 0;
            int slot_at[4];   // visit position -> slot
            {
                const int a0 = s_l ? 1 : 0, a1 = s_l ? 0 : 1, b0 = s_r ? 3 : 2, b1 = s_r ? 2 : 3;
                if (s_root) slot_at[0] = b0, slot_at[1] = b1, slot_at[2] = a0, slot_at[3] = a1;
                else slot_at[0] = a0, slot_at[1] = a1, slot_at[2] = b0, slot_at[3] = b1;
            }
            const unsigned qlo[3] = {w[8], w[10], w[12]}, qhi[3] = {w[9], w[11], w[13]};
            float te[4], tx[4];
            bool pass[4];
            int ref[4];
            for (int k = 0; k < 4; ++k) {
                const int sl = slot_at[k];
                float tmin = 0, tmax = 0;
                for (int a = 0; a < 3; ++a) {
                    float lo = fmaf((float)((qlo[a] >> (8 * sl)) & 255u), scale[a], f[a]);
                    float hi = fmaf((float)((qhi[a] >> (8 * sl)) & 255u), scale[a], f[a]);
                    if (g_walk_exact_slots && !S.wide_slot_node.empty() && S.wide_slot_node[(size_t)cur * 4 + sl] >= 0) {
                        lo = S.nodes[(size_t)S.wide_slot_node[(size_t)cur * 4 + sl]].bmin[a];
                        hi = S.nodes[(size_t)S.wide_slot_node[(size_t)cur * 4 + sl]].bmax[a];
                    }
                    const float t_in = ((neg[a] ? hi : lo) - oo[a]) * ii[a], t_out = ((neg[a] ? lo : hi) - oo[a]) * ii[a];
                    // fmaxf / fminf: of a number and a NaN the number, as the device's max3 / min3
                    tmin = a == 0 ? t_in : fmaxf(tmin, t_in);
                    tmax = a == 0 ? t_out : fminf(tmax, t_out);
                }
                ref[k] = (int)w[4 + sl];
                te[k] = tmin, tx[k] = tmax;
                pass[k] = tmin <= tmax && tmax > 0.0f && tmin < h.t;   // an empty slot: inverted box; its ref is POP
            }
            int first = -1;
            for (int k = 0; k < 4; ++k)
                if (pass[k]) {
                    first = k;
                    break;
                }
            for (int k = 3; k > first && first >= 0; --k)
                if (pass[k] && top < 64) { stack[top++] = {ref[k], te[k]}; ++g_wide_pushes; if (top > g_wide_maxsp) g_wide_maxsp = top; }
            cur = first >= 0 ? ref[first] : POP;
        }
    }

    // ---- camera (camera::get_samples, src/camera.cpp:117-224) ---------------------------------
    int sample_count() const {
        if (cam.num_samples == 1) return 1;
        int g = (int)sqrt((double)cam.num_samples);
        return g * g;
    }
    void camera_sample(int i, int j, int s, uint32_t key, V &o, V &dir, float &time) {
        V pos = mk(cam.position), cu = mk(cam.u), cv = mk(cam.v), cw = mk(cam.w), center = mk(cam.center);
        if (cam.num_samples == 1) {
            float su = cam.near_left + (cam.near_right - cam.near_left) * (i + 0.5f) / cam.resx;
            float sv = cam.near_bottom + (cam.near_top - cam.near_bottom) * (j + 0.5f) / cam.resy;
            V target = add(add(center, scl(cu, su)), scl(cv, sv));
            dir = normalize(sub(target, pos));
            o = pos;
            time = rnd(key, DIM_CAM_TIME);
            return;
        }
        int g = (int)sqrt((double)cam.num_samples);
        int sy = s / g, sx = s % g;
        float r1 = rnd(key, DIM_CAM_R1), r2 = rnd(key, DIM_CAM_R2);
        float sub_u = (sx + r1) / g;
        float sub_v = (sy + r2) / g;
        float su = cam.near_left + (cam.near_right - cam.near_left) * (i + sub_u) / cam.resx;
        float sv = cam.near_bottom + (cam.near_top - cam.near_bottom) * (j + sub_v) / cam.resy;
        V target = add(add(center, scl(cu, su)), scl(cv, sv));
        dir = normalize(sub(target, pos));
        o = pos;
        if (cam.aperture_size > 0.0f && cam.focus_distance > 0.0f) {
            float cos_theta = -dot(dir, cw);
            if (fabsf(cos_theta) < 1e-6f) cos_theta = 1e-6f;
            float t = cam.focus_distance / cos_theta;
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

    // ---- shape queries ---------------------------------------------------------------------
    const rt_material &material_of(const Hit &h) const {
        const int ntri = (int)S.triangles.size(), nsph = (int)S.spheres.size();
        if (h.shape < ntri) return S.materials[S.instances[h.instance].material];
        if (h.shape < ntri + nsph) return S.materials[S.spheres[h.shape - ntri].material];
        return S.materials[S.planes[h.shape - ntri - nsph].material];
    }
    void textures_of(const Hit &h, int &off, int &count) const {
        const int ntri = (int)S.triangles.size(), nsph = (int)S.spheres.size();
        if (h.shape < ntri) {
            off = S.instances[h.instance].tex_offset;
            count = S.instances[h.instance].tex_count;
        } else if (h.shape < ntri + nsph) {
            off = S.spheres[h.shape - ntri].tex_offset;
            count = S.spheres[h.shape - ntri].tex_count;
        } else {
            off = S.planes[h.shape - ntri - nsph].tex_offset;
            count = S.planes[h.shape - ntri - nsph].tex_count;
        }
    }
    V emission_of(const Hit &h) const {   // getEmission, src/shapes.cpp:318-325, src/shapes.h:183
        const int ntri = (int)S.triangles.size(), nsph = (int)S.spheres.size();
        if (h.shape < ntri) return mk(S.instances[h.instance].emission);
        if (h.shape < ntri + nsph) return mk(S.spheres[h.shape - ntri].emission);
        return mk(0, 0, 0);
    }
    // get_normal: triangle src/shapes.cpp:152-197, sphere :390-405, plane :504
    V normal_of(const Hit &h, V p) const {
        const int ntri = (int)S.triangles.size(), nsph = (int)S.spheres.size();
        if (h.shape < ntri) {
            const rt_triangle &t = S.triangles[h.shape];
            const rt_instance &in = S.instances[h.instance];
            if (!t.smooth) return normalize(xdir(in.normal, S.accel[h.shape].flat_normal));
            V po = xpoint(in.inv_model, p);
            V c1 = mk(t.c1);
            V v0 = sub(mk(t.c2), c1), v1 = sub(mk(t.c3), c1), v2 = sub(po, c1);
            float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
            const float denom = d00 * d11 - d01 * d01;
            float v = (d11 * d20 - d01 * d21) / denom;
            float w = (d00 * d21 - d01 * d20) / denom;
            float u = 1.0f - v - w;
            V n = add(add(scl(mk(t.n1), u), scl(mk(t.n2), v)), scl(mk(t.n3), w));
            n = normalize(n);
            return normalize(xdir(in.normal, n));
        }
        if (h.shape < ntri + nsph) {
            const rt_sphere &s = S.spheres[h.shape - ntri];
            V po = xpoint(s.inv_model, p);
            return normalize(xdir(s.normal, sub(po, mk(s.center))));
        }
        return mk(S.planes[h.shape - ntri - nsph].normal);
    }
    // calculate_uv: triangle src/shapes.cpp:87-122, sphere :449-465, plane :529-548
    void uv_of(const Hit &h, V p, float &u, float &v) const {
        const int ntri = (int)S.triangles.size(), nsph = (int)S.spheres.size();
        if (h.shape < ntri) {
            const rt_triangle &t = S.triangles[h.shape];
            const rt_instance &in = S.instances[h.instance];
            V local = xpoint(in.inv_model, p);
            V c1 = mk(t.c1);
            V e1 = sub(mk(t.c2), c1), e2 = sub(mk(t.c3), c1), vp = sub(local, c1);
            float d00 = dot(e1, e1), d01 = dot(e1, e2), d11 = dot(e2, e2), d20 = dot(vp, e1), d21 = dot(vp, e2);
            float denom = d00 * d11 - d01 * d01;
            if (fabsf(denom) < 1e-10f) {
                u = 0.0f;
                v = 0.0f;
                return;
            }
            float inv = 1.0f / denom;
            float beta = (d11 * d20 - d01 * d21) * inv;
            float gamma = (d00 * d21 - d01 * d20) * inv;
            float alpha = 1.0f - beta - gamma;
            u = alpha * t.tex_u[0] + beta * t.tex_u[1] + gamma * t.tex_u[2];
            v = alpha * t.tex_v[0] + beta * t.tex_v[1] + gamma * t.tex_v[2];
            return;
        }
        if (h.shape < ntri + nsph) {
            const rt_sphere &s = S.spheres[h.shape - ntri];
            V q = normalize(sub(xpoint(s.inv_model, p), mk(s.center)));
            float phi = om::atan2f_(q.z, q.x);
            float theta = om::asinf_(q.y);
            const float PI = 3.14159265359f;
            u = 1.0f - (phi + PI) / (2.0f * PI);
            v = 1.0f - (theta + PI / 2.0f) / PI;
            return;
        }
        const rt_plane &pl = S.planes[h.shape - ntri - nsph];
        V n = mk(pl.normal);
        V helper = fabsf(pl.raw_normal_x) > 0.9f ? mk(0, 1, 0) : mk(1, 0, 0);   // reads the parsed scalar, see rt_plane
        V tangent = normalize(cross(n, helper));
        V bitangent = cross(n, tangent);
        V to = sub(p, mk(pl.point));
        u = dot(to, tangent);
        v = dot(to, bitangent);
    }
    // get_tangent: triangle src/shapes.cpp:281-316, sphere :467-494, plane :550-562
    V tangent_of(const Hit &h, V p) const {
        const int ntri = (int)S.triangles.size(), nsph = (int)S.spheres.size();
        if (h.shape < ntri) {
            const rt_triangle &t = S.triangles[h.shape];
            const rt_instance &in = S.instances[h.instance];
            V c1 = mk(t.c1);
            V e1 = sub(mk(t.c2), c1), e2 = sub(mk(t.c3), c1);
            float du1 = t.tex_u[1] - t.tex_u[0], dv1 = t.tex_v[1] - t.tex_v[0];
            float du2 = t.tex_u[2] - t.tex_u[0], dv2 = t.tex_v[2] - t.tex_v[0];
            float det = (du1 * dv2 - du2 * dv1);
            V lt;
            if (fabsf(det) < 1e-8f) {
                lt = e1;
            } else {
                float f = 1.0f / det;
                lt = scl(sub(scl(e1, dv2), scl(e2, dv1)), f);
            }
            return normalize(xdir(in.model, lt));
        }
        if (h.shape < ntri + nsph) {
            // src/shapes.cpp:473-487 reads the scalar copies of a SIMD-only difference vector, which are still
            // zero, so the object-space tangent is always (1, 0, 0)
            const rt_sphere &s = S.spheres[h.shape - ntri];
            return normalize(xdir(s.model, mk(1, 0, 0)));
        }
        const rt_plane &pl = S.planes[h.shape - ntri - nsph];
        V helper = fabsf(pl.raw_normal_x) > 0.9f ? mk(0, 1, 0) : mk(1, 0, 0);
        return normalize(cross(mk(pl.normal), helper));
    }

    // ---- textures (src/texture.cpp) ------------------------------------------------------------
    // image::sample_nearest :64-93 / sample_bilinear :95-166; false = image missing, colour untouched
    bool image_sample(const rt_image &im, bool bilinear, float u, float v, V &color) {
        if (im.width <= 0 || im.height <= 0) return false;
        const int W = im.width, H = im.height;
        u -= floorf(u);
        v -= floorf(v);
        if (!bilinear) {
            int x = (int)(u * W), y = (int)(v * H);
            if (x >= W) x = W - 1;
            if (y >= H) y = H - 1;
            ++cnt.texels;
            if (im.is_hdr) {
                const float *p = &S.texels_f32[(size_t)im.offset + ((size_t)y * W + x) * 4];
                color = mk(p[0], p[1], p[2]);
            } else {
                const uint8_t *p = &S.texels_u8[(size_t)im.offset + ((size_t)y * W + x) * 3];
                color = mk(p[0] * (1.0f / 255.0f), p[1] * (1.0f / 255.0f), p[2] * (1.0f / 255.0f));
            }
            return true;
        }
        float px = u * W - 0.5f, py = v * H - 0.5f;
        int x = (int)floorf(px), y = (int)floorf(py);
        float ur = px - x, vr = py - y;
        float uo = 1.0f - ur, vo = 1.0f - vr;
        int x0 = (x % W + W) % W, x1 = ((x + 1) % W + W) % W;
        int y0 = (y % H + H) % H, y1 = ((y + 1) % H + H) % H;
        cnt.texels += 4;
        float out[3];
        for (int c = 0; c < 3; ++c) {
            float v00, v10, v01, v11;
            if (im.is_hdr) {
                const float *b = &S.texels_f32[(size_t)im.offset];
                v00 = b[((size_t)y0 * W + x0) * 4 + c];
                v10 = b[((size_t)y0 * W + x1) * 4 + c];
                v01 = b[((size_t)y1 * W + x0) * 4 + c];
                v11 = b[((size_t)y1 * W + x1) * 4 + c];
            } else {
                const uint8_t *b = &S.texels_u8[(size_t)im.offset];
                v00 = b[((size_t)y0 * W + x0) * 3 + c];
                v10 = b[((size_t)y0 * W + x1) * 3 + c];
                v01 = b[((size_t)y1 * W + x0) * 3 + c];
                v11 = b[((size_t)y1 * W + x1) * 3 + c];
            }
            float top = v00 * uo + v10 * ur;
            float bot = v01 * uo + v11 * ur;
            out[c] = top * vo + bot * vr;
        }
        if (im.is_hdr)
            color = mk(out[0], out[1], out[2]);
        else
            color = mk(out[0] * (1.0f / 255.0f), out[1] * (1.0f / 255.0f), out[2] * (1.0f / 255.0f));
        return true;
    }
    // Perlin::noise src/texture.cpp:204-235
    static float fade(float t) { return t * t * t * (t * (t * 6 - 15) + 10); }
    static float lerp(float t, float a, float b) { return a + t * (b - a); }
    static float grad(int hash, float x, float y, float z) {
        int h = hash & 15;
        float u = h < 8 ? x : y;
        float v = h < 4 ? y : (h == 12 || h == 14 ? x : z);
        return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
    }
    static float noise(float x, float y, float z) {
        static const uint8_t P[256] = {
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
        auto p = [&](int i) -> int { return P[i & 255]; };
        int X = (int)floorf(x) & 255, Y = (int)floorf(y) & 255, Z = (int)floorf(z) & 255;
        x -= floorf(x);
        y -= floorf(y);
        z -= floorf(z);
        float u = fade(x), v = fade(y), w = fade(z);
        int A = p(X) + Y, AA = p(A) + Z, AB = p(A + 1) + Z;
        int B = p(X + 1) + Y, BA = p(B) + Z, BB = p(B + 1) + Z;
        return lerp(w,
                    lerp(v, lerp(u, grad(p(AA), x, y, z), grad(p(BA), x - 1, y, z)),
                         lerp(u, grad(p(AB), x, y - 1, z), grad(p(BB), x - 1, y - 1, z))),
                    lerp(v, lerp(u, grad(p(AA + 1), x, y, z - 1), grad(p(BA + 1), x - 1, y, z - 1)),
                         lerp(u, grad(p(AB + 1), x, y - 1, z - 1), grad(p(BB + 1), x - 1, y - 1, z - 1))));
    }
    // perlin(), src/texture.cpp:241-274
    static float perlin(V h, const rt_texture &t) {
        float s = 0.0f;
        float x = h.x * t.noise_scale, y = h.y * t.noise_scale, z = h.z * t.noise_scale;
        float weight = 1.0f, max_amp = 0.0f;
        int K = (int)t.num_octaves;
        if (K < 1) K = 1;
        for (int k = 0; k < K; ++k) {
            s += weight * noise(x, y, z);
            max_amp += weight;
            x *= 2.0f;
            y *= 2.0f;
            weight *= 0.5f;
        }
        float sn = s / max_amp;
        if (t.noise_conversion == RT_NOISE_ABSVAL) return fabsf(sn);
        return (sn + 1.0f) * 0.5f;
    }
    // texture::sample, src/texture.cpp:168-197
    void tex_sample(const rt_texture &t, float u, float v, V h, V &color) {
        if (t.image >= 0) {
            image_sample(S.images[t.image], t.interp != RT_INTERP_NEAREST, u, v, color);
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
    static void latlong_uv(V d, float &u, float &v) {
        const float pi = 3.14159265359f;
        u = (1.0f + om::atan2f_(d.x, -d.z) / pi) * 0.5f;
        v = om::acosf_(d.y) / pi;
    }
    static void probe_uv(V d, float &u, float &v) {
        const float pi = 3.14159265359f;
        float len_sq = d.x * d.x + d.y * d.y;
        float r = 0.0f;
        if (len_sq > 1e-6f) r = (1.0f / pi) * om::acosf_(-d.z) / sqrtf(len_sq);
        u = (r * d.x + 1.0f) * 0.5f;
        v = (-r * d.y + 1.0f) * 0.5f;
    }
    // background lookup shared by the miss / depth-limit branches (src/ray_trace.cpp:649-679, 691-721, 887-903)
    V background(V dir, bool whitted, float pixelu, float pixelv) {
        if (S.d.background_texture >= 0) {
            const rt_texture &bg = S.textures[S.d.background_texture];
            if (bg.image >= 0) {
                V color = mk(0, 0, 0);
                if (!whitted || cam.is_hdr) {
                    float u, v;
                    if (S.d.is_probe)
                        probe_uv(dir, u, v);
                    else
                        latlong_uv(dir, u, v);
                    tex_sample(bg, u, v, mk(0, 0, 0), color);
                } else {
                    tex_sample(bg, pixelu, pixelv, mk(pixelu, pixelv, 0), color);
                }
                return color;
            }
        }
        return mk(S.d.background_color);
    }

    // ---- directions (src/ray_trace.cpp:41-105, 408-492) ----------------------------------------
    static void onb(V normal, float helper_x, V &u, V &v, V &w) {
        w = normal;
        V helper = fabsf(helper_x) > 0.9f ? mk(0, 1, 0) : mk(1, 0, 0);
        u = normalize(cross(helper, w));
        v = normalize(cross(w, u));
    }
    static void sample_hemisphere(bool cosine, float r1, float r2, V normal, float helper_x, V &dir, float &pdf) {
        float phi = 2.0f * 3.14159265359f * r1;
        float sin_theta, cos_theta;
        if (cosine) {
            sin_theta = sqrtf(r2);
            cos_theta = sqrtf(1.0f - r2);
        } else {
            sin_theta = sqrtf(1.0f - r2 * r2);
            cos_theta = r2;
        }
        float x = sin_theta * om::cosf_(phi), y = sin_theta * om::sinf_(phi), z = cos_theta;
        V u, v, w;
        onb(normal, helper_x, u, v, w);
        dir = normalize(add(add(scl(u, x), scl(v, y)), scl(w, z)));
        pdf = cosine ? cos_theta / 3.14159265359f : 1.0f / (2.0f * 3.14159265359f);
    }
    static void add_roughness(V &R, float roughness, float r_xi1, float r_xi2) {   // :408-447
        float ax = fabsf(R.x), ay = fabsf(R.y), az = fabsf(R.z);
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
    static V reflect_dir(V N, V I, float roughness, uint32_t key) {   // :449-462
        float d = dot(I, N);
        V R = add(I, scl(N, d * -2.0f));
        if (roughness > 0.0f) add_roughness(R, roughness, rnd(key, DIM_REFLECT_XI1), rnd(key, DIM_REFLECT_XI2));
        return normalize(R);
    }
    static bool refract_dir(V N, V I, float n1, float n2, float roughness, uint32_t key, V &T) {   // :464-492
        float cosi = -dot(I, N);
        float eta = n1 / n2;
        float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
        if (k < 0.0f) return false;
        T = add(scl(I, eta), scl(N, (eta * cosi - sqrtf(k))));
        if (roughness > 0.0f) add_roughness(T, roughness, rnd(key, DIM_REFRACT_XI1), rnd(key, DIM_REFRACT_XI2));
        T = normalize(T);
        return true;
    }
    // exact conductor Fresnel term, src/ray_trace.cpp:751-786 (== :952-987)
    static float conductor_fresnel(float cos_theta, float n, float k) {
        float c = cos_theta, c2 = c * c, s2 = 1.0f - c2;
        float n2 = n * n, k2 = k * k, n2_k2 = n2 - k2, n2k2_4 = 4.0f * n2 * k2;
        float a2_b2 = sqrtf((n2_k2 - s2) * (n2_k2 - s2) + n2k2_4);
        float a2 = 0.5f * (a2_b2 + n2_k2 - s2);
        float a = sqrtf(a2);
        float b2 = a2_b2 - a2;
        float b = sqrtf(b2);
        float rs_num = a2_b2 + c2 - (2.0f * a * c);
        float rs_den = a2_b2 + c2 + (2.0f * a * c);
        float rs = (rs_den == 0.0f) ? 1.0f : rs_num / rs_den;
        float ac = a * c, bc = b * c;
        float t1n = ac - s2, t1d = ac + s2;
        float rp_num = (t1n * t1n) + (bc * bc);
        float rp_den = (t1d * t1d) + (bc * bc);
        float rp = (rp_den == 0.0f) ? 1.0f : rs * (rp_num / rp_den);
        float F = 0.5f * (rs + rp);
        return F < 0.0f ? 0.0f : (F > 1.0f ? 1.0f : F);   // std::clamp
    }

    // ---- normal / bump mapping (src/ray_trace.cpp:494-644) --------------------------------------
    void apply_normal_map(const Hit &h, V p, int toff, int tcount, V &normal) {
        for (int k = 0; k < tcount; ++k) {
            const rt_texture &tex = S.textures[S.texture_refs[toff + k]];
            if (tex.decal_mode != RT_DECAL_REPLACE_NORMAL) continue;
            float u, v;
            uv_of(h, p, u, v);
            V raw = mk(0.5f, 0.5f, 1.0f);
            tex_sample(tex, u, v, p, raw);
            V ts = sub(scl(raw, 2.0f), mk(1.0f, 1.0f, 1.0f));
            V T = tangent_of(h, p);
            float ndott = dot(normal, T);
            T = normalize(sub(T, scl(normal, ndott)));
            V B = cross(normal, T);
            V nn = add(add(scl(T, ts.x), scl(B, ts.y)), scl(normal, ts.z));
            normal = normalize(nn);
            return;
        }
    }
    void apply_bump_map(const Hit &h, V p, int toff, int tcount, V &normal) {
        for (int k = 0; k < tcount; ++k) {
            const rt_texture &tex = S.textures[S.texture_refs[toff + k]];
            if (tex.decal_mode != RT_DECAL_BUMP_NORMAL) continue;
            if (tex.image >= 0) {
                const rt_image &im = S.images[tex.image];
                float u, v;
                uv_of(h, p, u, v);
                float du = 1.0f / im.width, dv = 1.0f / im.height;
                V hc = mk(0, 0, 0), hu = mk(0, 0, 0), hv = mk(0, 0, 0);
                tex_sample(tex, u, v, p, hc);
                tex_sample(tex, u + du, v, p, hu);
                tex_sample(tex, u, v + dv, p, hv);
                float dh_du = (hu.x - hc.x) * tex.bump_factor;
                float dh_dv = (hv.x - hc.x) * tex.bump_factor;
                V T = tangent_of(h, p);
                float ndott = dot(normal, T);
                T = normalize(sub(T, scl(normal, ndott)));
                V B = cross(normal, T);
                V nn = sub(sub(normal, scl(T, dh_du)), scl(B, dh_dv));
                normal = normalize(nn);
            } else {
                const float eps = 0.01f;
                const float inv_eps = 1.0f / eps;
                V hc, hx, hy, hz;
                tex_sample(tex, 0, 0, p, hc);
                tex_sample(tex, 0, 0, add(p, mk(eps, 0, 0)), hx);
                tex_sample(tex, 0, 0, add(p, mk(0, eps, 0)), hy);
                tex_sample(tex, 0, 0, add(p, mk(0, 0, eps)), hz);
                V g = mk((hx.x - hc.x) * inv_eps, (hy.x - hc.x) * inv_eps, (hz.x - hc.x) * inv_eps);
                float gdn = dot(g, normal);
                V surf = sub(g, scl(normal, gdn));
                normal = normalize(sub(normal, scl(surf, tex.bump_factor)));
            }
            return;
        }
    }

    // resolve_textures, src/ray_trace.cpp:107-152.  kd_rr is what Russian roulette later reads through
    // vec3's scalar copies: the SIMD-only blend_kd update leaves them at the material's value.
    void resolve_textures(const Hit &h, V p, const rt_material &m, int toff, int tcount, V &kd, V &ks, V &kd_rr, bool &replace_all,
                          V &replace_color) {
        kd = mk(m.diffuse);
        ks = mk(m.specular);
        kd_rr = kd;
        replace_all = false;
        for (int k = 0; k < tcount; ++k) {
            const rt_texture &tex = S.textures[S.texture_refs[toff + k]];
            float u, v;
            if (tex.decal_mode == RT_DECAL_REPLACE_KD) {
                uv_of(h, p, u, v);
                tex_sample(tex, u, v, p, kd);
                kd_rr = kd;
            } else if (tex.decal_mode == RT_DECAL_BLEND_KD) {
                uv_of(h, p, u, v);
                V tmp = mk(0, 0, 0);
                tex_sample(tex, u, v, p, tmp);
                kd = scl(add(mk(m.diffuse), tmp), 0.5f);
            } else if (tex.decal_mode == RT_DECAL_REPLACE_KS) {
                uv_of(h, p, u, v);
                tex_sample(tex, u, v, p, ks);
            } else if (tex.decal_mode == RT_DECAL_REPLACE_ALL) {
                uv_of(h, p, u, v);
                replace_color = mk(0, 0, 0);
                tex_sample(tex, u, v, p, replace_color);
                replace_all = true;
                return;
            }
        }
    }

    // ---- lights (Light::get_sample, src/lights.cpp) ----------------------------------------------
    void light_sample(const rt_light &L, V p, V normal, float ru, float rv, V &radiance, V &ldir, float &dist) {
        const float pi = 3.14159265359f;
        switch (L.type) {
            case RT_LIGHT_POINT: {   // :8-40
                V du = sub(mk(L.position), p);
                float d2 = dot(du, du);
                dist = sqrtf(d2);
                if (d2 < 1e-6f) {
                    ldir = du;
                    radiance = mk(0, 0, 0);
                    return;
                }
                ldir = scl(du, 1.0f / dist);
                radiance = scl(mk(L.color), 1.0f / d2);
                return;
            }
            case RT_LIGHT_AREA: {   // :60-109
                float uo = (ru - 0.5f) * L.size, vo = (rv - 0.5f) * L.size;
                V sp = add(add(mk(L.position), scl(mk(L.u), uo)), scl(mk(L.v), vo));
                V du = sub(sp, p);
                float d2 = dot(du, du);
                dist = sqrtf(d2);
                ldir = scl(du, 1.0f / dist);
                float cos_l = dot(mk(L.normal), scl(ldir, -1.0f));
                if (cos_l <= 0.0f) {
                    radiance = mk(0, 0, 0);
                    return;
                }
                float area = L.size * L.size;
                float factor = (area * cos_l) / d2;
                radiance = scl(mk(L.color), factor);
                return;
            }
            case RT_LIGHT_DIRECTIONAL: {   // :120-141
                dist = FLT_MAX;
                ldir = scl(mk(L.direction), -1.0f);
                radiance = mk(L.color);
                return;
            }
            case RT_LIGHT_SPOT: {   // :158-212
                V du = sub(mk(L.position), p);
                float d2 = dot(du, du);
                dist = sqrtf(d2);
                if (d2 < 1e-6f) {
                    ldir = du;
                    radiance = mk(0, 0, 0);
                    return;
                }
                ldir = scl(du, 1.0f / dist);
                float cos_alpha = dot(scl(ldir, -1.0f), mk(L.direction));
                float spot = 0.0f;
                if (cos_alpha > L.cos_falloff) {
                    spot = 1.0f;
                } else if (cos_alpha > L.cos_coverage) {
                    float ratio = (cos_alpha - L.cos_coverage) / (L.cos_falloff - L.cos_coverage);
                    spot = ratio * ratio * ratio * ratio;
                }
                radiance = scl(mk(L.color), spot / d2);
                return;
            }
            case RT_LIGHT_ENV: {   // :219-316
                dist = FLT_MAX;
                float phi = 2.0f * pi * ru, cos_t, sin_t, pdf;
                if (L.cosine_sampling) {
                    cos_t = sqrtf(rv);
                    sin_t = sqrtf(1.0f - cos_t * cos_t);
                    pdf = cos_t / pi;
                } else {
                    cos_t = rv;
                    sin_t = sqrtf(1.0f - cos_t * cos_t);
                    pdf = 1.0f / (2.0f * pi);
                }
                float lx = sin_t * om::cosf_(phi), ly = sin_t * om::sinf_(phi), lz = cos_t;
                V w = normal;
                V up = fabsf(w.y) < 0.9f ? mk(0, 1, 0) : mk(0, 0, 1);
                V u = normalize(cross(up, w));
                V vb = normalize(cross(w, u));
                ldir = normalize(add(add(scl(u, lx), scl(vb, ly)), scl(w, lz)));
                float uu, vv;
                if (L.is_probe)
                    probe_uv(ldir, uu, vv);
                else
                    latlong_uv(ldir, uu, vv);
                radiance = mk(0, 0, 0);
                if (L.image >= 0) image_sample(S.images[L.image], true, uu, vv, radiance);
                if (pdf < 1e-6f) pdf = 1e-6f;
                radiance = scl(radiance, 1.0f / pdf);
                return;
            }
            case RT_LIGHT_TRIANGLE: {   // :340-386
                float sr = sqrtf(ru);
                float u = 1.0f - sr, v = rv * sr, w = 1.0f - u - v;
                V sp = add(add(scl(mk(L.v0), w), scl(mk(L.v1), u)), scl(mk(L.v2), v));
                V du = sub(sp, p);
                float d2 = dot(du, du);
                dist = sqrtf(d2);
                ldir = scl(du, 1.0f / dist);
                float cos_l = fabsf(dot(mk(L.normal), scl(ldir, -1.0f)));
                radiance = scl(mk(L.color), (L.area * cos_l) / d2);
                return;
            }
            case RT_LIGHT_SPHERE: {   // :393-461
                const rt_sphere &s = S.spheres[L.sphere];
                float phi = 2.0f * 3.14159265359f * ru;
                float cos_t = 1.0f - 2.0f * rv;
                float sin_t = sqrtf(fmax_(0.0f, 1.0f - cos_t * cos_t));
                V unit = mk(sin_t * om::cosf_(phi), sin_t * om::sinf_(phi), cos_t);
                // src/lights.cpp:418-422: the sampled surface point exists only in SIMD lanes; mult_vec reloads its
                // operand from the (still zero) scalar copies, so the position used is model * (0,0,0,1)
                V sp = xpoint(s.model, mk(0, 0, 0));
                V nl = normalize(xdir(s.normal, unit));
                V du = sub(sp, p);
                float d2 = dot(du, du);
                dist = sqrtf(d2);
                ldir = scl(du, 1.0f / dist);
                float cos_l = dot(nl, scl(ldir, -1.0f));
                if (cos_l <= 0.0f) {
                    radiance = mk(0, 0, 0);
                    return;
                }
                radiance = scl(mk(L.color), (L.area * cos_l) / d2);
                return;
            }
        }
        radiance = mk(0, 0, 0);
        ldir = mk(0, 0, 0);
        dist = 0.f;
    }

    // calculate_color, src/ray_trace.cpp:154-406.  `color` is accumulated into, as in the reference.
    void calculate_color(V normal, const rt_material &m, int toff, int tcount, V p, V ray_origin, const Hit &h, float time,
                         bool is_hdr, bool include_ambient, uint32_t key, V &color) {
        V kd, ks, kd_rr, rep;
        bool replace_all;
        resolve_textures(h, p, m, toff, tcount, kd, ks, kd_rr, replace_all, rep);
        if (replace_all) {
            color = rep;
            return;
        }
        if (include_ambient) color = add(color, mul(mk(m.ambient), mk(S.d.ambient_light)));
        V view = normalize(sub(ray_origin, p));
        const float seps = S.d.shadow_ray_epsilon;
        uint32_t k = 0;
        for (const rt_light &L : S.lights) {
            int ns = L.samples;
            float weight = 1.0f / ns;
            for (int s = 0; s < ns; ++s, ++k) {
                float r1 = rnd(key, DIM_LIGHT + 2 * k), r2 = rnd(key, DIM_LIGHT + 2 * k + 1);
                V rad, ldir;
                float dist;
                light_sample(L, p, normal, r1, r2, rad, ldir, dist);
                if (rad.x == 0 && rad.y == 0 && rad.z == 0) continue;
                V so = add(p, scl(ldir, seps));
                ++cnt.shadow;
                Hit sh = intersect(so, ldir, time, false, seps, true, dist);
                if (sh.hit && sh.t < dist - seps) continue;
                float ndotl = fmax_(dot(normal, ldir), 0.0f);
                V hv = normalize(add(ldir, view));
                V fd = mk(0, 0, 0), fs = mk(0, 0, 0);
                const float p_exp = m.phong_exponent;
                const float PI = 3.14159265359f;
                switch (m.brdf) {
                    case RT_BRDF_ORIGINAL_BLINN_PHONG: {
                        fd = scl(kd, ndotl);
                        float ndoth = fmax_(dot(normal, hv), 0.0f);
                        fs = scl(ks, om::powf_(ndoth, p_exp));
                        break;
                    }
                    case RT_BRDF_ORIGINAL_PHONG: {
                        fd = scl(kd, ndotl);
                        V R = reflect_dir(normal, scl(ldir, -1.0f), 0.f, 0);
                        float rdoth = fmax_(dot(R, view), 0.0f);
                        fs = scl(ks, om::powf_(rdoth, p_exp));
                        break;
                    }
                    case RT_BRDF_MODIFIED_BLINN_PHONG: {
                        float ndoth = fmax_(dot(normal, hv), 0.0f);
                        if (m.normalized) {
                            fd = scl(kd, 1.0f / PI * ndotl);
                            float norm = (p_exp + 8.0f) / (8.0f * PI);
                            fs = scl(ks, norm * om::powf_(ndoth, p_exp) * ndotl);
                        } else {
                            fd = scl(kd, ndotl);
                            fs = scl(ks, om::powf_(ndoth, p_exp) * ndotl);
                        }
                        break;
                    }
                    case RT_BRDF_MODIFIED_PHONG: {
                        V R = reflect_dir(normal, scl(ldir, -1.0f), 0.f, 0);
                        float rdoth = fmax_(dot(R, view), 0.0f);
                        if (m.normalized) {
                            fd = scl(kd, 1.0f / PI * ndotl);
                            float norm = (p_exp + 2.0f) / (2.0f * PI);
                            fs = scl(ks, norm * om::powf_(rdoth, p_exp) * ndotl);
                        } else {
                            fd = scl(kd, ndotl);
                            fs = scl(ks, om::powf_(rdoth, p_exp) * ndotl);
                        }
                        break;
                    }
                    case RT_BRDF_TORRANCE_SPARROW: {
                        float ndoth = fmax_(dot(normal, hv), 0.0f);
                        float ndotv = fmax_(dot(normal, view), 0.0f);
                        float vdoth = fmax_(dot(view, hv), 0.0f);
                        float D = (p_exp + 2.0f) / (2.0f * PI) * om::powf_(ndoth, p_exp);
                        float f_term = om::powf_(1.0f - vdoth, 5.0f);
                        V one = mk(1, 1, 1);
                        V F = add(ks, scl(sub(one, ks), f_term));
                        float G = 1.0f;
                        if (vdoth > 1e-6f) {
                            float g1 = (2.0f * ndoth * ndotv) / vdoth;
                            float g2 = (2.0f * ndoth * ndotl) / vdoth;
                            G = fmin_(1.0f, fmin_(g1, g2));
                        }
                        if (ndotv > 1e-6f) {
                            float denom = 4.0f * ndotv;
                            fs = scl(F, (D * G) / denom);
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
    }

    // ---- Whitted integrator, ray_tracer::trace_rec src/ray_trace.cpp:646-869 ----------------------
    V trace_rec(V o, V d, float time, bool culling, float pixelu, float pixelv, int depth, uint32_t key) {
        if ((float)depth >= S.d.max_depth) return background(d, true, pixelu, pixelv);
        Hit h = intersect(o, d, time, culling, S.d.intersection_epsilon, false, 1e30f);   // default stop_t, src/bvh.h:47
        if (!h.hit) return background(d, true, pixelu, pixelv);
        ++cnt.shaded;
        const rt_material &m = material_of(h);
        int toff, tcount;
        textures_of(h, toff, tcount);
        V p = add(o, scl(d, h.t));
        V normal = normal_of(h, p);
        apply_normal_map(h, p, toff, tcount, normal);
        apply_bump_map(h, p, toff, tcount, normal);
        const float seps = S.d.shadow_ray_epsilon;
        const bool is_hdr = cam.is_hdr != 0;
        V color = mk(0, 0, 0);
        if (m.type == RT_MAT_MIRROR) {
            V R = reflect_dir(normal, d, m.roughness, key);
            p = add(p, scl(normal, seps));
            ++cnt.secondary;
            color = trace_rec(p, R, time, culling, pixelu, pixelv, depth + 1, child_key(key, BRANCH_REFLECT));
            color = mul(color, mk(m.mirror));
            V own = mk(0, 0, 0);
            calculate_color(normal, m, toff, tcount, p, o, h, time, is_hdr, true, key, own);
            color = add(color, own);
        } else if (m.type == RT_MAT_CONDUCTOR) {
            V R = reflect_dir(normal, d, m.roughness, key);
            float Fs = conductor_fresnel(fabsf(dot(d, normal)), m.refraction_index, m.absorption_index);
            V F = mul(mk(Fs, Fs, Fs), mk(m.mirror));
            p = add(p, scl(normal, seps));
            ++cnt.secondary;
            V refl = trace_rec(p, R, time, culling, pixelu, pixelv, depth + 1, child_key(key, BRANCH_REFLECT));
            color = mul(refl, F);
            V own = mk(0, 0, 0);
            calculate_color(normal, m, toff, tcount, p, o, h, time, is_hdr, true, key, own);
            color = add(color, own);
        } else if (m.type == RT_MAT_DIELECTRIC) {
            float n1 = 1.0f, n2 = m.refraction_index;
            float cos_theta = dot(d, normal);
            bool entering = cos_theta < 0.0f;
            if (!entering) {
                std::swap(n1, n2);
                normal = scl(normal, -1.0f);
                cos_theta = -cos_theta;
            }
            float cos_i = -cos_theta;
            cos_i = cos_i < 0.0f ? 0.0f : (cos_i > 1.0f ? 1.0f : cos_i);
            float r0 = (n1 - n2) / (n1 + n2);
            r0 = r0 * r0;
            float F = r0 + (1.0f - r0) * om::powf_(1.0f - cos_i, 5.0f);
            V rdir = reflect_dir(normal, d, m.roughness, key);
            V ro = add(p, scl(normal, seps));
            ++cnt.secondary;
            V rc = trace_rec(ro, rdir, time, culling, pixelu, pixelv, depth + 1, child_key(key, BRANCH_REFLECT));
            rc = mul(rc, mk(m.mirror));
            V tdir, tc = mk(0, 0, 0);
            if (refract_dir(normal, d, n1, n2, m.roughness, key, tdir)) {
                V to = add(p, scl(normal, -seps));
                ++cnt.secondary;
                tc = trace_rec(to, tdir, time, false, pixelu, pixelv, depth + 1, child_key(key, BRANCH_REFRACT));
                if (!entering) {
                    V att = scl(mk(m.absorption_coefficient), -h.t);
                    att = mk(om::expf_(att.x), om::expf_(att.y), om::expf_(att.z));
                    tc = mul(tc, att);
                }
            }
            color = add(scl(rc, F), scl(tc, 1.0f - F));
            V own = mk(0, 0, 0);
            calculate_color(normal, m, toff, tcount, p, o, h, time, is_hdr, true, key, own);
            color = add(color, own);
        } else {
            calculate_color(normal, m, toff, tcount, p, o, h, time, is_hdr, true, key, color);
        }
        return color;
    }

    // ---- path tracer, ray_tracer::path_trace src/ray_trace.cpp:871-1139 ---------------------------
    // `diffuse_bounce`: the ray direction came out of sample_*_hemisphere, whose result lives only in SIMD lanes;
    // the environment lookup of a miss reads the direction's scalar copies (src/ray_trace.cpp:10-12), which are
    // still (0,0,0) for such rays, so every diffuse-bounce miss fetches the same texel: atan2(+0,-0) = pi gives
    // u = 1, acos(0)/pi gives v = 1/2 (lat-long); the probe mapping gives (1/2, 1/2).
    V path_trace(V o, V d, float time, bool culling, int depth, uint32_t key, bool diffuse_bounce = false) {
        V color = mk(0, 0, 0);
        Hit h = intersect(o, d, time, culling, S.d.intersection_epsilon, false, 1e30f);   // default stop_t, src/bvh.h:47
        if (!h.hit) {
            if (diffuse_bounce && S.d.background_texture >= 0 && S.textures[S.d.background_texture].image >= 0) {
                V c = mk(0, 0, 0);
                tex_sample(S.textures[S.d.background_texture], S.d.is_probe ? 0.5f : 1.0f, 0.5f, mk(0, 0, 0), c);
                return c;
            }
            return background(d, false, 0.f, 0.f);
        }
        ++cnt.shaded;
        const rt_material &m = material_of(h);
        color = add(color, emission_of(h));
        V p = add(o, scl(d, h.t));
        V normal = normal_of(h, p);
        // the scalar copy that get_orthonormal_basis reads (:45): the geometric normal, or for planes the parsed one
        const float helper_x = h.shape >= (int)(S.triangles.size() + S.spheres.size())
                                   ? S.planes[h.shape - (int)(S.triangles.size() + S.spheres.size())].raw_normal_x
                                   : normal.x;
        int toff, tcount;
        textures_of(h, toff, tcount);
        apply_normal_map(h, p, toff, tcount, normal);
        apply_bump_map(h, p, toff, tcount, normal);
        V kd, ks, kd_rr, rep;
        bool replace_all;
        resolve_textures(h, p, m, toff, tcount, kd, ks, kd_rr, replace_all, rep);
        if (replace_all) return rep;
        const float seps = S.d.shadow_ray_epsilon;
        if (m.type != RT_MAT_REGULAR) {
            if ((float)depth < S.d.max_depth) {
                if (m.type == RT_MAT_MIRROR) {
                    V R = reflect_dir(normal, d, m.roughness, key);
                    V no = add(p, scl(normal, seps));
                    ++cnt.secondary;
                    V rc = path_trace(no, R, time, culling, depth + 1, child_key(key, BRANCH_REFLECT));
                    color = mul(rc, mk(m.mirror));
                } else if (m.type == RT_MAT_CONDUCTOR) {
                    V R = reflect_dir(normal, d, m.roughness, key);
                    float Fs = conductor_fresnel(fabsf(dot(d, normal)), m.refraction_index, m.absorption_index);
                    V F = mul(mk(Fs, Fs, Fs), mk(m.mirror));
                    V no = add(p, scl(normal, seps));
                    ++cnt.secondary;
                    V rc = path_trace(no, R, time, culling, depth + 1, child_key(key, BRANCH_REFLECT));
                    color = mul(rc, F);
                } else {
                    float n1 = 1.0f, n2 = m.refraction_index;
                    float cos_theta = dot(d, normal);
                    bool entering = cos_theta < 0.0f;
                    V rn = normal;
                    if (!entering) {
                        std::swap(n1, n2);
                        rn = scl(normal, -1.0f);
                    } else {
                        cos_theta = -cos_theta;
                    }
                    float r0 = (n1 - n2) / (n1 + n2);
                    r0 = r0 * r0;
                    float F = r0 + (1.0f - r0) * om::powf_(1.0f - cos_theta, 5.0f);
                    V R = reflect_dir(rn, d, m.roughness, key);
                    V ro = add(p, scl(rn, seps));
                    ++cnt.secondary;
                    V rc = path_trace(ro, R, time, culling, depth + 1, child_key(key, BRANCH_REFLECT));
                    rc = mul(rc, mk(m.mirror));
                    V T;
                    if (refract_dir(rn, d, n1, n2, m.roughness, key, T)) {
                        V to = add(p, scl(rn, -seps));
                        ++cnt.secondary;
                        V tc = path_trace(to, T, time, culling, depth + 1, child_key(key, BRANCH_REFRACT));
                        if (!entering) {
                            V att = scl(mk(m.absorption_coefficient), -h.t);
                            att = mk(om::expf_(att.x), om::expf_(att.y), om::expf_(att.z));
                            tc = mul(tc, att);
                        }
                        color = add(scl(rc, F), scl(tc, 1.0f - F));
                    } else {
                        color = rc;
                    }
                }
            }
            return color;
        }
        float rr_prob = 1.0f;
        if (cam.russian_roulette && depth >= cam.min_recursion_depth) {
            float max_refl = fmax_(fmax_(kd_rr.x, kd_rr.y), kd_rr.z);
            rr_prob = fmin_(max_refl, 0.99f);
            if (rnd(key, DIM_RR) > rr_prob) return color;
        }
        if (cam.next_event_estimation) {
            V direct = mk(0, 0, 0);
            calculate_color(normal, m, toff, tcount, p, o, h, time, false, false, key, direct);
            color = add(color, direct);
        }
        if ((float)depth < S.d.max_depth) {
            V wi;
            float pdf;
            sample_hemisphere(cam.importance_sampling != 0, rnd(key, DIM_HEMI_1), rnd(key, DIM_HEMI_2), normal, helper_x, wi, pdf);
            int split = (depth == 0 && cam.splitting_factor > 1) ? cam.splitting_factor : 1;
            V accum = mk(0, 0, 0);
            for (int i = 0; i < split; ++i) {
                if (split > 1)
                    sample_hemisphere(cam.importance_sampling != 0, rnd(key, DIM_SPLIT + 2 * i), rnd(key, DIM_SPLIT + 2 * i + 1), normal,
                                      helper_x, wi, pdf);
                V no = add(p, scl(normal, seps));
                ++cnt.secondary;
                V bounce = path_trace(no, wi, time, culling, depth + 1, child_key(key, BRANCH_BOUNCE + i), true);
                float ndotl = fmax_(dot(normal, wi), 0.0f);
                if (pdf > 1e-6f) {
                    V fr = scl(kd, 1.0f / 3.14159265359f);
                    V val = scl(mul(bounce, fr), ndotl / pdf);
                    accum = add(accum, val);
                }
            }
            accum = scl(accum, 1.0f / split);
            if (rr_prob < 1.0f && rr_prob > 0.0f) accum = scl(accum, 1.0f / rr_prob);
            color = add(color, accum);
        }
        return color;
    }

    // process_pixel, src/main.cpp:451-484, with ray_tracer::trace src/ray_trace.cpp:1141-1172
    void pixel(int i, int j, float *out) {
        float pixelv = 1.0f - ((float)j / (float)(cam.resy - 1));
        float pixelu = (float)i / (float)(cam.resx - 1);
        const int n = sample_count();
        const uint32_t pixel_index = (uint32_t)j * (uint32_t)cam.resx + (uint32_t)i;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int s = 0; s < n; ++s) {
            uint32_t key = sample_key(cam.seed, pixel_index, (uint32_t)s);
            V o, d;
            float time;
            camera_sample(i, j, s, key, o, d, time);
            ++cnt.primary;
            uint32_t root = child_key(key, BRANCH_ROOT);
            V c = cam.path_tracing ? path_trace(o, d, time, false, 0, root) : trace_rec(o, d, time, false, pixelu, pixelv, 0, root);
            if (cam.sample_max_val > 0.0f) {
                c.x = fmin_(c.x, cam.sample_max_val);
                c.y = fmin_(c.y, cam.sample_max_val);
                c.z = fmin_(c.z, cam.sample_max_val);
            }
            acc[0] += c.x;
            acc[1] += c.y;
            acc[2] += c.z;
        }
        float ns = (float)n;
        out[0] = acc[0] / ns;
        out[1] = acc[1] / ns;
        out[2] = acc[2] / ns;
    }
};

void fill_stats(const Counters &c, double ms, rt_stats *st) {
    if (!st) return;
    memset(st, 0, sizeof *st);
    st->primary_rays = c.primary;
    st->shadow_rays = c.shadow;
    st->secondary_rays = c.secondary;
    st->nodes_visited = c.nodes;
    st->triangle_tests = c.tri;
    st->sphere_tests = c.sph;
    st->plane_tests = c.pln;
    st->shaded_hits = c.shaded;
    st->texel_fetches = c.texels;
    st->kernel_ms = (float)ms;
    st->total_ms = (float)ms;
    st->shadow_rays_traced = c.shadow;   // the reference traverses once per light sample
}

template <class T> void copy_array(std::vector<T> &dst, const T *src, int64_t n) {
    dst.assign(src, src + (src ? n : 0));
}

}  // namespace

extern "C" {

int orc_scene_create(const rt_scene_desc *d, orc_scene **out) {
    if (!d || !out || d->abi_version != RT_ABI_VERSION) return 1;
    orc_scene *s = new orc_scene();
    s->d = *d;
    copy_array(s->triangles, d->triangles, d->n_triangles);
    copy_array(s->instances, d->instances, d->n_instances);
    copy_array(s->spheres, d->spheres, d->n_spheres);
    copy_array(s->planes, d->planes, d->n_planes);
    copy_array(s->materials, d->materials, d->n_materials);
    copy_array(s->textures, d->textures, d->n_textures);
    copy_array(s->texture_refs, d->texture_refs, d->n_texture_refs);
    copy_array(s->images, d->images, d->n_images);
    copy_array(s->texels_u8, d->texels_u8, d->n_texels_u8);
    copy_array(s->texels_f32, d->texels_f32, d->n_texels_f32);
    copy_array(s->lights, d->lights, d->n_lights);
    copy_array(s->nodes, d->bvh_nodes, d->n_bvh_nodes);
    copy_array(s->prims, d->bvh_prims, d->n_bvh_prims);
    s->accel.resize(s->triangles.size());
    for (size_t i = 0; i < s->triangles.size(); ++i) {   // src/shapes.cpp:19-48
        const rt_triangle &t = s->triangles[i];
        V c1s = scl(mk(t.c1), 1000.0f), c2s = scl(mk(t.c2), 1000.0f), c3s = scl(mk(t.c3), 1000.0f);
        TriAccel &a = s->accel[i];
        a.c1s = c1s;
        a.e1 = sub(c2s, c1s);
        a.e2 = sub(c3s, c1s);
        a.flat_normal = normalize(cross(a.e1, a.e2));
    }
    *out = s;
    return 0;
}

void orc_scene_destroy(orc_scene *s) { delete s; }

// Renders rows [0, resy) with `threads` host threads pulling scan lines from an atomic counter
// (the reference's row_dynamic policy, src/main.cpp:488-497).
int orc_render(orc_scene *s, const rt_camera_desc *cam, float *hdr_out, rt_stats *stats, int threads) {
    if (!s || !cam || !hdr_out) return 1;
    if (threads < 1) threads = 1;
    auto t0 = std::chrono::high_resolution_clock::now();
    std::atomic<int> next(0);
    std::vector<Counters> counters(threads);
    auto work = [&](int tid) {
        Tracer tr(*s, *cam);
        while (true) {
            int j = next++;
            if (j >= cam->resy) break;
            for (int i = 0; i < cam->resx; ++i) tr.pixel(i, j, hdr_out + ((size_t)j * cam->resx + i) * 3);
        }
        counters[tid] = tr.cnt;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    Counters total;
    for (auto &c : counters) total.add(c);
    auto t1 = std::chrono::high_resolution_clock::now();
    fill_stats(total, std::chrono::duration<double, std::milli>(t1 - t0).count(), stats);
    return 0;
}

// Primary visibility of each pixel's first sample: shape index, reference instance id, distance.
int orc_trace_primary(orc_scene *s, const rt_camera_desc *cam, int32_t *shape_out, int32_t *id_out, float *t_out, rt_stats *stats) {
    if (!s || !cam) return 1;
    Tracer tr(*s, *cam);
    const int ntri = (int)s->triangles.size();
    for (int j = 0; j < cam->resy; ++j)
        for (int i = 0; i < cam->resx; ++i) {
            size_t px = (size_t)j * cam->resx + i;
            uint32_t key = sample_key(cam->seed, (uint32_t)px, 0);
            V o, d;
            float time;
            tr.camera_sample(i, j, 0, key, o, d, time);
            ++tr.cnt.primary;
            Hit h = tr.intersect(o, d, 0.0f, false, s->d.intersection_epsilon, false, 1e30f);
            shape_out[px] = h.hit ? h.shape : -1;
            id_out[px] = !h.hit ? -1 : (h.shape < ntri ? s->instances[h.instance].ref_id : (h.shape < ntri + (int)s->spheres.size() ? 123456 : -1));
            t_out[px] = h.hit ? h.t : 1e30f;
        }
    fill_stats(tr.cnt, 0.0, stats);
    return 0;
}

// Single-ray probe for known-answer tests of bvh::intersect.
int orc_intersect(orc_scene *s, const float *o, const float *d, float time, int culling, float eps, int any_hit, float stop_t,
                  int32_t *shape, int32_t *instance, float *t) {
    rt_camera_desc cam{};
    Tracer tr(*s, cam);
    Hit h = tr.intersect(mk(o), mk(d), time, culling != 0, eps, any_hit != 0, stop_t);
    *shape = h.hit ? h.shape : -1;
    *instance = h.hit ? h.instance : -1;
    *t = h.t;
    return h.hit ? 1 : 0;
}

// The same probe through the product's wide tree; returns -1 when the scene's tree does not qualify for one.
int orc_intersect_wide(orc_scene *s, const float *o, const float *d, float time, int culling, float eps, int any_hit, float stop_t,
                       int32_t *shape, int32_t *instance, float *t) {
    if (s->wide_state == 0) {
        int root = 0, n_top = 0;
        s->leaf_node_at.assign(s->prims.size(), -1);
        for (size_t i = 0; i < s->nodes.size(); ++i)
            if (s->nodes[i].n_prims > 0) s->leaf_node_at[(size_t)s->nodes[i].offset] = (int)i;
        s->wide_state = rt::build_wide_nodes(s->nodes.data(), (int)s->nodes.size(), 127, 64, s->wide, root, n_top) ? 1 : -1;
    }
    if (s->wide_state < 0) return -1;
    rt_camera_desc cam{};
    Tracer tr(*s, cam);
    Hit h = tr.intersect_wide(mk(o), mk(d), time, culling != 0, eps, any_hit != 0, stop_t);
    *shape = h.hit ? h.shape : -1;
    *instance = h.hit ? h.instance : -1;
    *t = h.t;
    return h.hit ? 1 : 0;
}

// 1: orc_render / orc_trace_primary walk the wide tree of scenes whose wide tree has been built (orc_intersect_wide
// builds it); counters then count the root box only.  For checking that both walks give the same image.
void orc_set_wide(int on) { g_walk_wide = on; }
void orc_set_exact_slots(int on) { g_walk_exact_slots = on; }
long orc_walk_mismatches(void) { return g_walk_mismatches.load(); }
// work of the walks on the calling thread since the last call (render with one thread): wide steps, leaves reached by
// the wide walk, pushes, deepest stack, interior nodes expanded by the two-child walk, leaves whose exact box passed
void orc_walk_work(long *out) {
    const long v[6] = {g_wide_steps, g_wide_leaf_arrivals, g_wide_pushes, g_wide_maxsp, g_two_interior, g_wide_leaf_visits};
    memcpy(out, v, sizeof v);
    g_wide_steps = g_wide_leaf_arrivals = g_wide_pushes = g_wide_maxsp = g_two_interior = g_wide_leaf_visits = 0;
}

// Checks every record of the wide tree against the two-child tree it was built from: each slot's decoded box
// (fmaf(q, step, origin), the device's arithmetic) contains the exact box of the node it stands for, empty slots are
// inverted, interior slots lead to the record of that node, leaf slots carry the leaf's entry range, and every leaf is
// reachable exactly once.  Returns the number of violations (0 = sound), -1 when the scene has no wide tree.
static int ensure_wide(orc_scene *s) {
    if (s->wide_state == 0) {
        float o[3] = {0, 0, 0}, d[3] = {0, 0, 1}, t;
        int32_t a, b;
        orc_intersect_wide(s, o, d, 0.f, 0, 1e-6f, 0, 1e30f, &a, &b, &t);
    }
    return s->wide_state;
}
long orc_wide_check(orc_scene *s, long *n_records) {
    if (ensure_wide(s) < 0) return -1;
    *n_records = (long)s->wide.size();
    long bad = 0, leaves_seen = 0, leaves = 0;
    for (const rt_bvh_node &n : s->nodes) leaves += n.n_prims > 0;
    // walk records and two-child nodes together
    std::vector<std::pair<int, int>> todo{{0, 0}};   // (record, node)
    std::vector<char> seen(s->wide.size(), 0);
    while (!todo.empty()) {
        const auto [rec, node] = todo.back();
        todo.pop_back();
        if (rec < 0 || rec >= (int)s->wide.size() || seen[(size_t)rec]++) {
            ++bad;
            continue;
        }
        const rt_bvh_node *N = s->nodes.data();
        const int L = node + 1, R = N[node].offset;
        int slot[4] = {N[L].n_prims ? L : L + 1, N[L].n_prims ? -1 : N[L].offset, N[R].n_prims ? R : R + 1, N[R].n_prims ? -1 : N[R].offset};
        if (s->wide_slot_node.size() != s->wide.size() * 4) s->wide_slot_node.assign(s->wide.size() * 4, -1);
        for (int k = 0; k < 4; ++k) s->wide_slot_node[(size_t)rec * 4 + k] = slot[k];
        unsigned w[16];
        const float *f = s->wide[(size_t)rec].f;
        memcpy(w, f, sizeof w);
        const unsigned sx = w[3] & 0x7f800000u;
        float step[3];
        memcpy(&step[0], &sx, 4);
        step[1] = f[14], step[2] = f[15];
        const unsigned axes = w[3] & 0x1ffu;
        const unsigned want_axes = (1u << N[node].axis) | (N[L].n_prims ? 0u : 1u << (3 + N[L].axis)) | (N[R].n_prims ? 0u : 1u << (6 + N[R].axis));
        if (axes != want_axes) ++bad;
        for (int k = 0; k < 4; ++k) {
            const int ref = (int)w[4 + k];
            for (int a = 0; a < 3; ++a) {
                const float lo = fmaf((float)((w[8 + 2 * a] >> (8 * k)) & 255u), step[a], f[a]);
                const float hi = fmaf((float)((w[9 + 2 * a] >> (8 * k)) & 255u), step[a], f[a]);
                if (slot[k] < 0) {
                    if (!(lo > hi)) ++bad;
                } else if (!(lo <= N[slot[k]].bmin[a] && hi >= N[slot[k]].bmax[a])) {
                    ++bad;
                }
            }
            if (slot[k] < 0) {
                if (ref != rt::WIDE_REF_NONE) ++bad;
            } else if (N[slot[k]].n_prims > 0) {
                ++leaves_seen;
                if ((unsigned)ref != (0x80000000u | (unsigned)N[slot[k]].offset)) ++bad;
            } else {
                todo.push_back({ref, slot[k]});
            }
        }
    }
    if (leaves_seen != leaves) ++bad;
    for (char c : seen) bad += c != 1;
    return bad;
}

// Copies the wide records (16 words each) to `out` (room for `n_records`); returns their number.
long orc_wide_records(orc_scene *s, float *out, long n_records) {
    if (ensure_wide(s) < 0) return -1;
    const long n = std::min<long>(n_records, (long)s->wide.size());
    if (out) memcpy(out, s->wide.data(), (size_t)n * sizeof(rt::WideRecord));
    return (long)s->wide.size();
}

// The product's builder on a caller-made two-child tree (for trees an ingest would not produce): number of records,
// -1 when the builder declines the tree.
long orc_wide_build_raw(const rt_bvh_node *nodes, int n_nodes, int stack_entries) {
    std::vector<rt::WideRecord> out;
    int root = 0, n_top = 0;
    return rt::build_wide_nodes(nodes, n_nodes, 63, stack_entries, out, root, n_top) ? (long)out.size() : -1;
}

// Builds the wide tree again and returns the milliseconds the product's builder took (host-side cost of a scene).
double orc_wide_build_ms(orc_scene *s) {
    std::vector<rt::WideRecord> out;
    int root = 0, n_top = 0;
    const auto t0 = std::chrono::high_resolution_clock::now();
    rt::build_wide_nodes(s->nodes.data(), (int)s->nodes.size(), 63, 64, out, root, n_top);
    return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
}

// Diagnostic for layout studies: walks a render's rays through the wide tree (single thread) and returns what each
// ray touched; `out` receives up to `room` entries (records >= 0, leaves -1 - first entry, -2^30 ends a ray).
long orc_wide_log(orc_scene *s, const rt_camera_desc *cam, int32_t *out, long room) {
    if (ensure_wide(s) < 0) return -1;
    std::vector<int> log;
    std::vector<float> img((size_t)cam->resx * cam->resy * 3);
    rt_stats st;
    g_wide_log = &log;
    g_walk_wide = 3;
    orc_render(s, cam, img.data(), &st, 1);
    g_walk_wide = 0;
    g_wide_log = nullptr;
    const long n = std::min<long>(room, (long)log.size());
    memcpy(out, log.data(), (size_t)n * 4);
    return n;
}

// Many rays at once: o, d are n x 3; mode 0 the two-child walk, 1 the wide walk.  Results per ray: shape (-1 none),
// instance, distance.
int orc_intersect_many(orc_scene *s, int mode, int n, const float *o, const float *d, const float *stop_t, int any_hit, float eps,
                       int32_t *shape, int32_t *instance, float *t) {
    for (int i = 0; i < n; ++i) {
        const int keep = g_walk_wide;
        g_walk_wide = 0;
        int rc = mode ? orc_intersect_wide(s, o + 3 * i, d + 3 * i, 0.f, 0, eps, any_hit, stop_t[i], shape + i, instance + i, t + i)
                      : orc_intersect(s, o + 3 * i, d + 3 * i, 0.f, 0, eps, any_hit, stop_t[i], shape + i, instance + i, t + i);
        g_walk_wide = keep;
        if (rc < 0) return rc;
    }
    return 0;
}

// Probes of the shared arithmetic definitions, used to check the device code bit for bit.
float orc_math(int fn, float a, float b) {
    switch (fn) {
        case 0: return om::sinf_(a);
        case 1: return om::cosf_(a);
        case 2: return om::asinf_(a);
        case 3: return om::acosf_(a);
        case 4: return om::atan2f_(a, b);
        case 5: return om::expf_(a);
        case 6: return om::logf_(a);
        case 7: return om::powf_(a, b);
        case 8: return a / b;
        case 9: return sqrtf(a);
        case 10: return rnd(om::f2u(a), om::f2u(b));
        default: return 0.f;
    }
}
uint32_t orc_sample_key(uint32_t seed, uint32_t pixel, uint32_t sample) { return sample_key(seed, pixel, sample); }
uint32_t orc_child_key(uint32_t key, uint32_t branch) { return child_key(key, branch); }
float orc_rnd(uint32_t key, uint32_t dim) { return rnd(key, dim); }

}  // extern "C"
