// Device-side scene access of the rendering core: BVH traversal (closest / any hit), primitive tests, surface
// queries, textures, lights, direction sampling and the camera.  Every function cites the reference routine it
// restates; the arithmetic order is part of the definition (see rt_math.h).
#pragma once
#include <hip/hip_runtime.h>

#include "gpu_render.h"
#include "rt_math.h"

namespace rt {

constexpr int TILE = 32;              // image tile edge used for sharding

enum {
    DIM_REFLECT_XI1 = 0, DIM_REFLECT_XI2 = 1, DIM_REFRACT_XI1 = 2, DIM_REFRACT_XI2 = 3, DIM_RR = 4, DIM_HEMI_1 = 5, DIM_HEMI_2 = 6,
    DIM_SPLIT = 8, DIM_LIGHT = 4096,
    DIM_CAM_R1 = 0, DIM_CAM_R2 = 1, DIM_CAM_R3 = 2, DIM_CAM_R4 = 3, DIM_CAM_TIME = 4,
    BRANCH_ROOT = 0, BRANCH_REFLECT = 1, BRANCH_REFRACT = 2, BRANCH_BOUNCE = 3
};

// ---- device-resident scene -------------------------------------------------------------------
// nodes:  4 x float4 per INTERIOR BVH node, holding the boxes of both children:
//         {Lmin.xyz, Lmax.x} {Lmax.yz, Rmin.xy} {Rmin.z, Rmax.xyz} {bits(Lref), bits(Rref), bits(1 << split axis), 0}
//         child ref >= 0: interior node index; < 0: leaf = 1<<31 | first leaf entry (the leaf's last entry says so itself)
// prims:  4 x float4 (one 64-byte block) per leaf entry, in leaf order; the last 24 bytes hold the box of the leaf the
//         entry belongs to, so that the wide walk's exact leaf test and the first entry's test read the same block:
//         triangle {c1s.xyz, e1.x} {e1.yz, e2.xy} {e2.z, bits(entry word), leaf min.xy} {leaf min.z, leaf max.xyz}
//         sphere   {...unused...}                 {0, bits(entry word), leaf min.xy}    {leaf min.z, leaf max.xyz}
//         entry word: bit 31 sphere, bit 30 the placement moves or transforms the ray, bit 29 last entry of its leaf, bits 0..28
//         placement (triangles) or sphere index.  A leaf holds as many entries as the tree gives it (the reference's count is
//         16 bits wide, src/bvh.h:20)
// prim_meta: 2 x int4 per leaf entry: what shading needs of a hit entry, in one 32-byte block:
//         {bits(instance) as above, shape, reference id, smooth shading?} {flat normal.xyz, 0}
// inst:   4 x float4 per placement {inv row0} {inv row1} {inv row2} {motionblur.xyz, bits(flags)}
//         flags bit0 = identity transform, bit1 = has motion blur
// wnodes: 4 x float4 per WIDE node: an interior node merged with those of its two children that are interior, so
//         that it holds the boxes of up to four descendants ("slots": 0, 1 from the left child, 2, 3 from the right),
//         each box quantised to 8 bits per coordinate on a grid local to the node (never smaller than the box):
//         {origin.xyz, bits(2^ex | axis masks)} {bits(slot refs 0..3)} {lo.x, hi.x, lo.y, hi.y: one byte per slot}
//         {lo.z, hi.z, bits(2^ey), bits(2^ez)};  axis masks: 1 << split axis of the node | of its left child << 3 | of
//         its right child << 6;  slot ref: wide node index, leaf reference as above, or REF_POP for an empty slot
struct DevScene {
    const float4 *nodes;
    const float4 *wnodes;   // null when the scene has no wide tree
    const float4 *prims;
    const int4 *prim_meta;
    const float4 *inst;
    const rt_instance *instances;
    const rt_triangle *triangles;
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
    int root_ref;
    int n_top;   // the first n_top node records are the top of the tree in breadth-first order (kept in LDS by the trace kernel)
    int wroot_ref, n_wtop;   // the same for the wide tree
    float root_min[3], root_max[3];
    float ieps, seps, max_depth;
    float id_scale;   // k of the placements whose inverse is k I (see plain_placement)
    float ambient[3], bg[3];
    int bg_tex, is_probe;
};

// A placement whose model matrix is the identity has the inverse k I, where k is what the scene's inverse routine makes
// of 1 / 1: the reference's approximate reciprocal and Newton step give 1 - 2^-24 (src/algebra.cpp:345-347), and the ray
// its triangle test sees is the world ray times k (src/shapes.cpp:224-230) -- one last bit that decides whether a shadow
// ray towards a sample on an emitter's triangle ends before or behind distance - epsilon.  Such a placement without
// motion blur is "plain": its leaf entries carry no transform bit, and the test multiplies by the scene's one k
// (DevScene::id_scale) instead of reading a matrix.
__host__ __device__ inline bool plain_placement(const rt_instance &in, float k) {
    if (!in.identity || in.motionblur[0] != 0.f || in.motionblur[1] != 0.f || in.motionblur[2] != 0.f) return false;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
            if (in.inv_model[4 * r + c] != (r == c ? k : 0.f)) return false;
    return true;
}
__host__ __device__ inline bool scaled_identity(const rt_instance &in, float k) {   // the same with motion blur allowed
    if (!in.identity) return false;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
            if (in.inv_model[4 * r + c] != (r == c ? k : 0.f)) return false;
    return true;
}

struct Counters {   // per-lane, reduced per wave at kernel end
    uint32_t primary, shadow, secondary, shaded;
    uint32_t nodes, tri, sph, pln, texels;
    uint32_t traced;   // shadow rays actually traversed (light samples with identical rays share one)
};

struct Hit {
    float t;
    int prim;   // >= 0 leaf entry; <= -2 plane (-2 - index); -1 none
};

// ---- traversal ---------------------------------------------------------------------------------
// bvh::intersect (src/bvh.cpp:383-480) with intersect_aabb_fast (:348-381): planes first, then an ordered
// stack traversal (near child first by the sign of the ray direction on the split axis).  A hit replaces
// the current one only if t > eps and t < t_hit, so of equal-distance hits the one found first is kept.
//
// Device formulation: an interior node record holds the boxes of BOTH children, so one 64-byte fetch
// decides both; the far child is pushed together with its entry distance and re-checked against the
// current closest hit when popped.  That is the reference's "test the node's own box after popping" with
// the t_hit-independent part of the test done one step earlier: the same nodes pass, in the same order,
// and the same primitives are tested in the same order.  The first LDS_STACK entries of a lane's stack
// live in LDS (one 8-byte column per lane), deeper entries spill to private memory.
//
// The traversal is cut into steps (ray set-up, pop, interior-node step, one primitive test) that the trace
// kernel schedules per wave; a lane's position is its `cur` reference:
//   0 <= cur < REF_POP   interior node to visit
//   cur < 0              leaf: 1<<31 | next leaf entry
//   REF_POP              take the next node from the stack
//   REF_IDLE             no ray
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 10
#endif
constexpr int LDS_STACK = RT_LDS_STACK;
constexpr int SPILL_STACK = 64 - LDS_STACK;   // LDS_STACK + SPILL_STACK = 64 = the reference's stack size (src/bvh.cpp:423)
constexpr int REF_POP = 0x40000000;
constexpr int REF_IDLE = 0x40000001;
constexpr int PRIM_STRIDE = 4;   // float4 per leaf entry: one 64-byte block
constexpr int PRIM_XFORM_BIT = 0x40000000;   // in a leaf entry's word (and the shading record's placement): the placement moves or transforms the ray
constexpr int PRIM_LAST_BIT = 0x20000000;    // in a leaf entry's word: the last entry of its leaf
constexpr int PRIM_INDEX_MASK = 0x1fffffff;  // placement or sphere index
constexpr int LEAF_ENTRY_MASK = 0x7fffffff;  // in a leaf reference: the entry

// Box test of intersect_aabb_fast without its final (t_min < t_hit) term; returns the entry distance.  The
// reference narrows [t_min, t_max] axis by axis and leaves as soon as the interval is empty; for ordered operands
// that is "largest entry <= smallest exit", evaluated here without branches.  (The two forms differ only when a
// product is NaN, 0 x inf: a ray parallel to a box face that starts exactly on it.)
__device__ __forceinline__ bool slab(float x0, float y0, float z0, float x1, float y1, float z1, float ox, float oy, float oz, float ix,
                                     float iy, float iz, bool nx, bool ny, bool nz, float &tmin_out) {
    const float tx0 = ((nx ? x1 : x0) - ox) * ix, tx1 = ((nx ? x0 : x1) - ox) * ix;
    const float ty0 = ((ny ? y1 : y0) - oy) * iy, ty1 = ((ny ? y0 : y1) - oy) * iy;
    const float tz0 = ((nz ? z1 : z0) - oz) * iz, tz1 = ((nz ? z0 : z1) - oz) * iz;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(tx0, ty0), tz0);
    const float tmax = __builtin_fminf(__builtin_fminf(tx1, ty1), tz1);
    tmin_out = tmin;
    return (tmin <= tmax) & (tmax > 0.0f);
}

// The same test for the two child boxes of a node at once: every subtraction and multiplication is a two-wide packed
// operation (L, R).  Element for element the arithmetic is that of slab().
__device__ __forceinline__ void slab_pair(f2 x0, f2 y0, f2 z0, f2 x1, f2 y1, f2 z1, float ox, float oy, float oz, float ix, float iy,
                                          float iz, bool nx, bool ny, bool nz, float t_hit, float &tl, float &tr, bool &pl, bool &pr) {
    const f2 tx0 = ((nx ? x1 : x0) - ox) * ix, tx1 = ((nx ? x0 : x1) - ox) * ix;
    const f2 ty0 = ((ny ? y1 : y0) - oy) * iy, ty1 = ((ny ? y0 : y1) - oy) * iy;
    const f2 tz0 = ((nz ? z1 : z0) - oz) * iz, tz1 = ((nz ? z0 : z1) - oz) * iz;
    tl = __builtin_fmaxf(__builtin_fmaxf(tx0.x, ty0.x), tz0.x);
    tr = __builtin_fmaxf(__builtin_fmaxf(tx0.y, ty0.y), tz0.y);
    const float ml = __builtin_fminf(__builtin_fminf(tx1.x, ty1.x), tz1.x);
    const float mr = __builtin_fminf(__builtin_fminf(tx1.y, ty1.y), tz1.y);
    pl = (tl <= ml) & (ml > 0.0f) & (tl < t_hit);
    pr = (tr <= mr) & (mr > 0.0f) & (tr < t_hit);
}

// What a lane keeps of its ray while it walks the tree.
struct RayState {
    float ox, oy, oz, dx, dy, dz;   // origin, direction (scalars: aggregates passed by value end up in private memory)
    float ix, iy, iz;   // 1 / d
    float time;
    float t;            // closest accepted distance so far (starts at the ray's stop distance)
    int prim;           // its leaf entry; <= -2 plane (-2 - index); -1 none
    int cur, sp;
    int sgn;            // bit a (and a + 3, a + 6) set: direction component a is negative
    bool culling;
};

// Ray set-up: the plane loop (plane::intersect, src/shapes.cpp:506-527) and the root box.  Returns false when the
// traversal is already over (any-hit satisfied by a plane, no BVH, or the root box is missed).
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool ray_begin(const DevScene &S, RayState &r, float eps, float stop_t, Counters &c) {
    r.t = stop_t;
    r.prim = -1;
    r.sp = 0;
    for (int pi = 0; pi < S.n_planes; ++pi) {
        if (COUNT) ++c.pln;
        const rt_plane &p = S.planes[pi];
        const V n = mk(p.normal);
        const float denom = dot(n, mk(r.dx, r.dy, r.dz));
        if (absf(denom) < eps) continue;
        const float num = dot(sub(mk(p.point), mk(r.ox, r.oy, r.oz)), n);
        const float tc = rt_div(num, denom);
        if (tc < eps) continue;
        if (tc > eps && tc < r.t) {
            r.t = tc;
            r.prim = -2 - pi;
            if (ANY) return false;
        }
    }
    if (S.n_nodes == 0) return false;
    r.ix = rt_div(1.0f, r.dx);
    r.iy = rt_div(1.0f, r.dy);
    r.iz = rt_div(1.0f, r.dz);
    r.sgn = ((r.ix < 0 ? 1 : 0) | (r.iy < 0 ? 2 : 0) | (r.iz < 0 ? 4 : 0)) * 0x49;   // three copies: bits 0-2, 3-5, 6-8 (see wide_step)
    if (COUNT) ++c.nodes;
    float tm;
    if (!slab(S.root_min[0], S.root_min[1], S.root_min[2], S.root_max[0], S.root_max[1], S.root_max[2], r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, r.ix < 0,
              r.iy < 0, r.iz < 0, tm) ||
        !(tm < r.t))
        return false;
    r.cur = S.root_ref;
    return true;
}

constexpr int STACK_STRIDE = 256;   // lanes per workgroup = entries per LDS stack level

// The private part of the stack is accessed as volatile words so that its loads and stores stay separate
// instructions from the LDS ones (merged, both would go through the slow generic address path).
// LEVELS: the entries of this stack that live in LDS (the any-hit kernel trades three of them for two more waves per SIMD).
template <int LEVELS = LDS_STACK>
__device__ __forceinline__ void stack_push(RayState &r, uint2 *lds, uint2 *spill, uint2 e) {
    if (r.sp < LEVELS) {
        lds[r.sp * STACK_STRIDE] = e;
    } else if (r.sp < 64) {
        volatile unsigned *w = (volatile unsigned *)(spill + (r.sp - LEVELS));
        w[0] = e.x;
        w[1] = e.y;
    } else {
        return;   // deeper than 64 would have overrun the reference's own stack; such entries are dropped
    }
    ++r.sp;
}

// One interior node: both child boxes, near child next, far child pushed.  `lds` is this lane's column of a
// __shared__ uint2 array laid out [LDS_STACK][STACK_STRIDE].
#ifndef RT_TOP_NODES
#define RT_TOP_NODES 31
#endif
constexpr int TOP_NODES = RT_TOP_NODES;   // node records a workgroup of the trace kernel keeps in LDS (0: none).  Measured with the
// two-box records, closest-hit time of a 4K / 64 spp frame: none 384 ms, 31: 379, 63: 375, 127 or more: 369.  LDS is what limits
// the workgroups per CU: stack (2 KB per level) plus these records (64 bytes each) stay under 26.6 KB for six workgroups; with
// the wide tree 11 levels + 63 records and 10 levels + 95 records measure the same (255 ms), 9 levels + 127 records 262 ms.
// Seven workgroups per CU (the closest-hit walk's default since) leave 22.8 KB each: 10 levels + 31 records.

// `top`: the workgroup's LDS copy of the first `n_top` node records (the top levels of the tree, which every ray walks);
// records read from it do not go through the vector cache, whose lookups are what bounds the traversal.
template <bool COUNT>
__device__ __forceinline__ void node_step(const DevScene &S, RayState &r, uint2 *lds, uint2 *spill, const float4 *top, int n_top, Counters &c) {
    float4 n0, n1, n2, n3;
    if (TOP_NODES > 0 && r.cur < n_top) {
        const float4 *np = top + 4 * r.cur;
        n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        asm volatile("" ::: );   // keeps these LDS reads apart from the global loads below (merged, they become flat loads)
    } else {
        const float4 *np = S.nodes + 4 * (size_t)r.cur;
        n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    }
    const bool nx = r.ix < 0, ny = r.iy < 0, nz = r.iz < 0;
    float tl, tr;
    bool pl, pr;
    slab_pair(f2{n0.x, n1.z}, f2{n0.y, n1.w}, f2{n0.z, n2.x}, f2{n0.w, n2.y}, f2{n1.x, n2.z}, f2{n1.y, n2.w}, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, nx,
              ny, nz, r.t, tl, tr, pl, pr);
    const bool neg = (r.sgn & __float_as_int(n3.z)) != 0;   // n3.z = 1 << split axis
    const int near_ref = __float_as_int(neg ? n3.y : n3.x), far_ref = __float_as_int(neg ? n3.x : n3.y);
    const bool pn = (neg & pr) | (!neg & pl), pf = (neg & pl) | (!neg & pr);
    if (COUNT) {
        ++c.nodes;             // the near child is popped (and its box tested) right after the push
        if (!pn) ++c.nodes;    // near failed: the far child is popped next
        // failing far children are kept so that their later pop is counted
        if (pn) stack_push(r, lds, spill, make_uint2((unsigned)far_ref, __float_as_uint(pf ? (neg ? tl : tr) : __int_as_float(0x7f800000))));
    } else {
        if (pn & pf) stack_push(r, lds, spill, make_uint2((unsigned)far_ref, __float_as_uint(neg ? tl : tr)));
    }
    r.cur = pn ? near_ref : (pf ? far_ref : REF_POP);
}

// One wide node (see DevScene::wnodes): up to four boxes at once, two levels of the tree per fetch of the same 64
// bytes.  The slots are visited in the order the two-level walk would visit them -- the side of the node's split the
// ray looks along first, and inside each side the same rule with that child's split axis -- so leaves are reached in
// the same relative order.  The quantised boxes contain the exact ones, and the arithmetic below is monotonic in the
// box coordinates, so whatever passes the exact test passes this one: the walk reaches a superset of the leaves the
// two-level walk visits; the trace kernel then tests each leaf's own exact box before its entries, which removes the
// surplus (see there), so both walks test the same entries in the same order and keep the same hit.  A stack
// entry's distance is the (never later) entry into the quantised box.
__device__ __forceinline__ float ubyte_f(unsigned v, int k) { return (float)((v >> (8 * k)) & 0xffu); }
__device__ __forceinline__ f2 fma2(f2 a, float b, float c) { return __builtin_elementwise_fma(a, f2{b, b}, f2{c, c}); }

// (the step proper, on a record the caller has fetched)
template <int LEVELS = LDS_STACK>
__device__ __forceinline__ void wide_step_on(RayState &r, uint2 *lds, uint2 *spill, float4 n0, float4 n1, float4 n2, float4 n3) {
    const unsigned head = __float_as_uint(n0.w);
    const float sx = __uint_as_float(head & 0x7f800000u), sy = n3.z, sz = n3.w;
    const unsigned along = (unsigned)r.sgn & head;
    const bool s_root = (along & 7u) != 0, s_l = (along & 0x38u) != 0, s_r = (along & 0x1c0u) != 0;
    // visit position -> slot, one byte each
    const unsigned lo16 = s_l ? 0x0001u : 0x0100u, hi16 = s_r ? 0x0203u : 0x0302u;
    const unsigned sel = s_root ? (lo16 << 16 | hi16) : (hi16 << 16 | lo16);
    const bool nx = r.ix < 0, ny = r.iy < 0, nz = r.iz < 0;
    const unsigned qlx = __float_as_uint(n2.x), qhx = __float_as_uint(n2.y), qly = __float_as_uint(n2.z), qhy = __float_as_uint(n2.w);
    const unsigned qlz = __float_as_uint(n3.x), qhz = __float_as_uint(n3.y);
    const unsigned ex = __builtin_amdgcn_perm(0, nx ? qhx : qlx, sel), fx = __builtin_amdgcn_perm(0, nx ? qlx : qhx, sel);
    const unsigned ey = __builtin_amdgcn_perm(0, ny ? qhy : qly, sel), fy = __builtin_amdgcn_perm(0, ny ? qly : qhy, sel);
    const unsigned ez = __builtin_amdgcn_perm(0, nz ? qhz : qlz, sel), fz = __builtin_amdgcn_perm(0, nz ? qlz : qhz, sel);
    // entry and exit distances of positions (0, 1) and (2, 3), per axis: ((origin + q * 2^e) - o) * (1 / d)
#define RT_WIDE_AXIS(q, s, p, o, inv, A, B)                                   \
    const f2 A = (fma2(f2{ubyte_f(q, 0), ubyte_f(q, 1)}, s, p) - o) * inv;   \
    const f2 B = (fma2(f2{ubyte_f(q, 2), ubyte_f(q, 3)}, s, p) - o) * inv;
    RT_WIDE_AXIS(ex, sx, n0.x, r.ox, r.ix, ex01, ex23)
    RT_WIDE_AXIS(fx, sx, n0.x, r.ox, r.ix, fx01, fx23)
    RT_WIDE_AXIS(ey, sy, n0.y, r.oy, r.iy, ey01, ey23)
    RT_WIDE_AXIS(fy, sy, n0.y, r.oy, r.iy, fy01, fy23)
    RT_WIDE_AXIS(ez, sz, n0.z, r.oz, r.iz, ez01, ez23)
    RT_WIDE_AXIS(fz, sz, n0.z, r.oz, r.iz, fz01, fz23)
#undef RT_WIDE_AXIS
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(ex01.x, ey01.x), ez01.x), m0 = __builtin_fminf(__builtin_fminf(fx01.x, fy01.x), fz01.x);
    const float t1 = __builtin_fmaxf(__builtin_fmaxf(ex01.y, ey01.y), ez01.y), m1 = __builtin_fminf(__builtin_fminf(fx01.y, fy01.y), fz01.y);
    const float t2 = __builtin_fmaxf(__builtin_fmaxf(ex23.x, ey23.x), ez23.x), m2 = __builtin_fminf(__builtin_fminf(fx23.x, fy23.x), fz23.x);
    const float t3 = __builtin_fmaxf(__builtin_fmaxf(ex23.y, ey23.y), ez23.y), m3 = __builtin_fminf(__builtin_fminf(fx23.y, fy23.y), fz23.y);
    // slot refs in visit order; an empty slot has an inverted box, and should arithmetic that is not a number let it
    // through, its ref is REF_POP, which only makes the lane pop
    const int c0 = __float_as_int(n1.x), c1 = __float_as_int(n1.y), c2 = __float_as_int(n1.z), c3 = __float_as_int(n1.w);
    const int e0 = s_l ? c1 : c0, e1 = s_l ? c0 : c1, e2 = s_r ? c3 : c2, e3 = s_r ? c2 : c3;
    const int d0 = s_root ? e2 : e0, d1 = s_root ? e3 : e1, d2 = s_root ? e0 : e2, d3 = s_root ? e1 : e3;
    const bool h0 = (t0 <= m0) & (m0 > 0.0f) & (t0 < r.t);
    const bool h1 = (t1 <= m1) & (m1 > 0.0f) & (t1 < r.t);
    const bool h2 = (t2 <= m2) & (m2 > 0.0f) & (t2 < r.t);
    const bool h3 = (t3 <= m3) & (m3 > 0.0f) & (t3 < r.t);
    // the first position that passes is next; the others wait on the stack, the earliest on top
    const bool p3 = h3 & (h0 | h1 | h2), p2 = h2 & (h0 | h1), p1 = h1 & h0;
    if (r.sp + 3 <= LEVELS) {   // all in LDS: store without branching, an entry above the top is never read
        int sp = r.sp;
        lds[sp * STACK_STRIDE] = make_uint2((unsigned)d3, __float_as_uint(t3));
        sp += p3 ? 1 : 0;
        lds[sp * STACK_STRIDE] = make_uint2((unsigned)d2, __float_as_uint(t2));
        sp += p2 ? 1 : 0;
        lds[sp * STACK_STRIDE] = make_uint2((unsigned)d1, __float_as_uint(t1));
        sp += p1 ? 1 : 0;
        r.sp = sp;
    } else {
        if (p3) stack_push<LEVELS>(r, lds, spill, make_uint2((unsigned)d3, __float_as_uint(t3)));
        if (p2) stack_push<LEVELS>(r, lds, spill, make_uint2((unsigned)d2, __float_as_uint(t2)));
        if (p1) stack_push<LEVELS>(r, lds, spill, make_uint2((unsigned)d1, __float_as_uint(t1)));
    }
    r.cur = h0 ? d0 : (h1 ? d1 : (h2 ? d2 : (h3 ? d3 : REF_POP)));
}

template <int LEVELS = LDS_STACK>
__device__ __forceinline__ void wide_step(const DevScene &S, RayState &r, uint2 *lds, uint2 *spill, const float4 *top, int n_top) {
    float4 n0, n1, n2, n3;
    if (TOP_NODES > 0 && r.cur < n_top) {
        const float4 *np = top + 4 * r.cur;
        n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        asm volatile("" ::: );   // keeps these LDS reads apart from the global loads below (merged, they become flat loads)
    } else {
        const float4 *np = S.wnodes + 4 * (size_t)r.cur;
        n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    }
    wide_step_on<LEVELS>(r, lds, spill, n0, n1, n2, n3);
}

// ---- the any-hit walk of the wide tree -------------------------------------------------------------------------------
// A ray that only asks whether anything is hit gets the same answer whatever the order of its walk: the answer is "some
// leaf whose own exact box the ray passes holds an entry the ray hits", and which leaves are reached does not depend on the
// order either (such a ray never shortens).  So its step leaves the slots in the order of the record -- no sign tests, no
// byte permutation, no reordering of the refs -- and its stack holds the references alone, 4 bytes an entry: whatever
// was pushed still qualifies when it is popped.  (4K / 64 spp frame, any hit: 102.4 -> 96.9 ms with the slots unsorted.)
template <int LEVELS>
__device__ __forceinline__ bool pop_any(RayState &r, const unsigned *lds, const unsigned *spill) {
    if (r.sp == 0) return false;
    --r.sp;
    if (r.sp < LEVELS) {
        r.cur = (int)lds[r.sp * STACK_STRIDE];
    } else {
        r.cur = (int)((const volatile unsigned *)spill)[r.sp - LEVELS];
    }
    return true;
}
// (as pop_inline: the top entry when it lies in LDS; true when the stack was empty)
template <int LEVELS>
__device__ __forceinline__ bool pop_any_inline(RayState &r, const unsigned *lds) {
    if (r.sp == 0) return true;
    if (r.sp <= LEVELS) {
        --r.sp;
        r.cur = (int)lds[r.sp * STACK_STRIDE];
    }
    return false;
}
template <int LEVELS>
__device__ __forceinline__ void wide_any_step_on(RayState &r, unsigned *lds, unsigned *spill, float4 n0, float4 n1, float4 n2, float4 n3) {
    const unsigned head = __float_as_uint(n0.w);
    const float sx = __uint_as_float(head & 0x7f800000u), sy = n3.z, sz = n3.w;
    const bool nx = r.ix < 0, ny = r.iy < 0, nz = r.iz < 0;
    const unsigned qlx = __float_as_uint(n2.x), qhx = __float_as_uint(n2.y), qly = __float_as_uint(n2.z), qhy = __float_as_uint(n2.w);
    const unsigned qlz = __float_as_uint(n3.x), qhz = __float_as_uint(n3.y);
    const unsigned ex = nx ? qhx : qlx, fx = nx ? qlx : qhx, ey = ny ? qhy : qly, fy = ny ? qly : qhy, ez = nz ? qhz : qlz, fz = nz ? qlz : qhz;
    // the arithmetic of wide_step, slot by slot: ((origin + q * 2^e) - o) * (1 / d)
#define RT_WIDE_AXIS(q, s, p, o, inv, A, B)                                   \
    const f2 A = (fma2(f2{ubyte_f(q, 0), ubyte_f(q, 1)}, s, p) - o) * inv;   \
    const f2 B = (fma2(f2{ubyte_f(q, 2), ubyte_f(q, 3)}, s, p) - o) * inv;
    RT_WIDE_AXIS(ex, sx, n0.x, r.ox, r.ix, ex01, ex23)
    RT_WIDE_AXIS(fx, sx, n0.x, r.ox, r.ix, fx01, fx23)
    RT_WIDE_AXIS(ey, sy, n0.y, r.oy, r.iy, ey01, ey23)
    RT_WIDE_AXIS(fy, sy, n0.y, r.oy, r.iy, fy01, fy23)
    RT_WIDE_AXIS(ez, sz, n0.z, r.oz, r.iz, ez01, ez23)
    RT_WIDE_AXIS(fz, sz, n0.z, r.oz, r.iz, fz01, fz23)
#undef RT_WIDE_AXIS
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(ex01.x, ey01.x), ez01.x), m0 = __builtin_fminf(__builtin_fminf(fx01.x, fy01.x), fz01.x);
    const float t1 = __builtin_fmaxf(__builtin_fmaxf(ex01.y, ey01.y), ez01.y), m1 = __builtin_fminf(__builtin_fminf(fx01.y, fy01.y), fz01.y);
    const float t2 = __builtin_fmaxf(__builtin_fmaxf(ex23.x, ey23.x), ez23.x), m2 = __builtin_fminf(__builtin_fminf(fx23.x, fy23.x), fz23.x);
    const float t3 = __builtin_fmaxf(__builtin_fmaxf(ex23.y, ey23.y), ez23.y), m3 = __builtin_fminf(__builtin_fminf(fx23.y, fy23.y), fz23.y);
    const int d0 = __float_as_int(n1.x), d1 = __float_as_int(n1.y), d2 = __float_as_int(n1.z), d3 = __float_as_int(n1.w);
    const bool h0 = (t0 <= m0) & (m0 > 0.0f) & (t0 < r.t);
    const bool h1 = (t1 <= m1) & (m1 > 0.0f) & (t1 < r.t);
    const bool h2 = (t2 <= m2) & (m2 > 0.0f) & (t2 < r.t);
    const bool h3 = (t3 <= m3) & (m3 > 0.0f) & (t3 < r.t);
    // the first slot that passes is next; the others wait on the stack
    const bool p3 = h3 & (h0 | h1 | h2), p2 = h2 & (h0 | h1), p1 = h1 & h0;
    if (r.sp + 3 <= LEVELS) {   // all in LDS: store without branching, an entry above the top is never read
        int sp = r.sp;
        lds[sp * STACK_STRIDE] = (unsigned)d3;
        sp += p3 ? 1 : 0;
        lds[sp * STACK_STRIDE] = (unsigned)d2;
        sp += p2 ? 1 : 0;
        lds[sp * STACK_STRIDE] = (unsigned)d1;
        sp += p1 ? 1 : 0;
        r.sp = sp;
    } else {
        const int refs[3] = {d3, d2, d1};
        const bool push[3] = {p3, p2, p1};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!push[k] || r.sp >= 64) continue;   // (deeper than 64 would have overrun the reference's own stack)
            if (r.sp < LEVELS)
                lds[r.sp * STACK_STRIDE] = (unsigned)refs[k];
            else
                ((volatile unsigned *)spill)[r.sp - LEVELS] = (unsigned)refs[k];
            ++r.sp;
        }
    }
    r.cur = h0 ? d0 : (h1 ? d1 : (h2 ? d2 : (h3 ? d3 : REF_POP)));
}

template <int LEVELS>
__device__ __forceinline__ void wide_any_step(const DevScene &S, RayState &r, unsigned *lds, unsigned *spill, const float4 *top, int n_top) {
    float4 n0, n1, n2, n3;
    if (r.cur < n_top) {
        const float4 *np = top + 4 * r.cur;
        n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        asm volatile("" ::: );   // (keeps the LDS reads apart from the global loads, see wide_step)
    } else {
        const float4 *np = S.wnodes + 4 * (size_t)r.cur;
        n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    }
    wide_any_step_on<LEVELS>(r, lds, spill, n0, n1, n2, n3);
}

// Pops until an entry whose box still starts before the closest hit; false when the stack is empty.  With
// CHECK false (an any-hit ray never shortens, so every pushed entry still qualifies) the first entry is taken.
template <bool CHECK, bool COUNT, int LEVELS = LDS_STACK>
__device__ __forceinline__ bool stack_pop(RayState &r, const uint2 *lds, const uint2 *spill, Counters &c) {
    while (r.sp > 0) {
        --r.sp;
        uint2 e;
        if (r.sp < LEVELS) {
            e = lds[r.sp * STACK_STRIDE];
        } else {
            const volatile unsigned *w = (const volatile unsigned *)(spill + (r.sp - LEVELS));
            e.x = w[0];
            e.y = w[1];
        }
        if (COUNT) ++c.nodes;
        if (!CHECK || __uint_as_float(e.y) < r.t) {
            r.cur = (int)e.x;
            return true;
        }
    }
    return false;
}

// One pop without the loop, for a lane that has just finished its node inside a step: takes the top entry when it
// lies in the LDS part of the stack; an entry that no longer qualifies is dropped and the lane stays at REF_POP, which
// the next round of stack_pop continues from -- the same entries are taken in the same order.  Returns true when the
// stack was empty: the ray is finished.
template <bool CHECK, int LEVELS = LDS_STACK>
__device__ __forceinline__ bool pop_inline(RayState &r, const uint2 *lds) {
    if (r.sp == 0) return true;
    if (r.sp <= LEVELS) {
        const uint2 e = lds[(r.sp - 1) * STACK_STRIDE];
        --r.sp;
        if (!CHECK || __uint_as_float(e.y) < r.t) r.cur = (int)e.x;
    }
    return false;
}

// Leaf entry k against the ray; true with the distance in `tc` when the primitive reports a hit (t > eps).
template <bool COUNT>
// p0, p1, p2: the first three float4 of the entry (loaded by the caller, together with the leaf's box where it needs it)
__device__ __forceinline__ bool prim_test(const DevScene &S, float4 p0, float4 p1, float4 p2, const RayState &r, float eps, float &tc, Counters &c) {
    const int inst = __float_as_int(p2.y);
    if (inst >= 0) {   // triangle::intersect, src/shapes.cpp:212-279
        if (COUNT) ++c.tri;
        const V ro = mk(r.ox, r.oy, r.oz), rd = mk(r.dx, r.dy, r.dz);
        V oo, dd;
        if (!(inst & PRIM_XFORM_BIT)) {   // a plain placement: the inverse is id_scale times the identity
            oo = scl(ro, S.id_scale);
            dd = scl(rd, S.id_scale);
        } else {
            const float4 *ip = S.inst + 4 * (size_t)(inst & PRIM_INDEX_MASK);
            const float4 r3 = ip[3];
            const int flags = __float_as_int(r3.w);
            V moved = ro;
            if (flags & 2) moved = sub(ro, scl(mk(r3.x, r3.y, r3.z), r.time));
            if (flags & 1) {
                oo = scl(moved, S.id_scale);
                dd = scl(rd, S.id_scale);
            } else {
                const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
                oo = mk((r0.x * moved.x + r0.y * moved.y) + (r0.z * moved.z + r0.w), (r1.x * moved.x + r1.y * moved.y) + (r1.z * moved.z + r1.w),
                        (r2.x * moved.x + r2.y * moved.y) + (r2.z * moved.z + r2.w));
                dd = mk((r0.x * rd.x + r0.y * rd.y) + r0.z * rd.z, (r1.x * rd.x + r1.y * rd.y) + r1.z * rd.z,
                        (r2.x * rd.x + r2.y * rd.y) + r2.z * rd.z);
            }
        }
        const V c1s = mk(p0.x, p0.y, p0.z), e1 = mk(p0.w, p1.x, p1.y), e2 = mk(p1.z, p1.w, p2.x);
        const V os = scl(oo, 1000.0f);
        const V hh = cross(dd, e2);
        const float det = dot(e1, hh);
        if (r.culling) {
            if (det < eps) return false;
        } else {
            if (absf(det) < eps) return false;
        }
        const float f = rt_div(1.0f, det);
        const V s = sub(os, c1s);
        float u = dot(s, hh);
        u *= f;
        if (u < (0.0f - eps) || u > (1.0f + eps)) return false;
        const V q = cross(s, e1);
        float v = dot(dd, q);
        v *= f;
        if (v < (0.0f - eps) || (u + v) > (1.0f + eps)) return false;
        tc = dot(e2, q);
        tc *= f;
        tc *= (1.0f / 1000.0f);
        return tc > eps;
    }
    // sphere::intersect, src/shapes.cpp:407-447
    if (COUNT) ++c.sph;
    const rt_sphere &sp_ = S.spheres[inst & PRIM_INDEX_MASK];
    const V so = xpoint(sp_.inv_model, mk(r.ox, r.oy, r.oz));
    const V sd = xdir(sp_.inv_model, mk(r.dx, r.dy, r.dz));
    const V oc = sub(so, mk(sp_.center));
    const float a = dot(sd, sd);
    float b = dot(oc, sd);
    b *= 2.0f;
    float cc = dot(oc, oc);
    cc -= (sp_.radius * sp_.radius);
    const float disc = b * b - 4 * a * cc;
    if (disc < 0.0f) return false;
    const float sq = rt_sqrt(disc);
    const float inv2a = rt_div(1.0f, 2.0f * a);
    const float t1 = (-b - sq) * inv2a, t2 = (-b + sq) * inv2a;
    if (t1 > eps)
        tc = t1;
    else if (t2 > eps)
        tc = t2;
    else
        return false;
    return true;
}

// ---- hit decoding -----------------------------------------------------------------------------
struct Surface {
    int kind;       // 0 triangle, 1 sphere, 2 plane
    int index;      // triangle / sphere / plane index
    int instance;   // placement (triangles)
    int material;
    int toff, tcount;
    bool smooth;    // triangles: interpolated vertex normals
    V flat;         // triangles: the face normal in object space
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
    const int4 meta = S.prim_meta[2 * (size_t)h.prim], meta1 = S.prim_meta[2 * (size_t)h.prim + 1];
    s.smooth = meta.w != 0;
    s.flat = mk(__int_as_float(meta1.x), __int_as_float(meta1.y), __int_as_float(meta1.z));
    int inst = meta.x;
    if (inst >= 0) {
        inst &= ~PRIM_XFORM_BIT;
        s.kind = 0;
        s.index = meta.y;
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
        const rt_instance &in = S.instances[s.instance];
        if (!s.smooth) return normalize(xdir(in.normal, s.flat));
        const rt_triangle &t = S.triangles[s.index];
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
// SphereLight::get_sample (src/lights.cpp:393-461) in two parts.  The reference transforms the scalar copies of
// the sampled point, which are still zero (:418-422): the position used is model * (0,0,0,1), so direction and
// distance depend on the shading point only, and the random numbers only turn the emitter normal.
__device__ __forceinline__ void sphere_light_geometry(const rt_sphere &s, V p, V &ldir, float &dist, float &d2) {
    const V sp = xpoint(s.model, mk(0, 0, 0));
    const V du = sub(sp, p);
    d2 = dot(du, du);
    dist = rt_sqrt(d2);
    ldir = scl(du, rt_div(1.0f, dist));
}
__device__ __forceinline__ V sphere_light_radiance(const rt_light &L, const rt_sphere &s, V ldir, float d2, float ru, float rv) {
    const float phi = 2.0f * 3.14159265359f * ru;
    const float cos_t = 1.0f - 2.0f * rv;
    const float sin_t = rt_sqrt(max_std(0.0f, 1.0f - cos_t * cos_t));
    const V unit = mk(sin_t * rt_cos(phi), sin_t * rt_sin(phi), cos_t);
    const V nl = normalize(xdir(s.normal, unit));
    const float cos_l = dot(nl, scl(ldir, -1.0f));
    if (cos_l <= 0.0f) return mk(0, 0, 0);
    return scl(mk(L.color), rt_div(L.area * cos_l, d2));
}

// What a sphere emitter's samples need of the light and its sphere: the same for every lane of a wave, read once per light.
struct SphereEmitter {
    float n[9];        // the sphere's normal matrix, rows 0..2, columns 0..2
    float color[3];
    float area;
};
__device__ __forceinline__ float wave_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ SphereEmitter sphere_emitter(const rt_light &L, const rt_sphere &s) {
    SphereEmitter e;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) e.n[3 * r + c] = wave_uniform(s.normal[4 * r + c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) e.color[c] = wave_uniform(L.color[c]);
    e.area = wave_uniform(L.area);
    return e;
}
// sphere_light_radiance for two samples at once: element for element the operations of the function above, the
// multiplications and additions as two-wide packed instructions.  Returns the radiance per colour channel, {sample a,
// sample b}.
__device__ __forceinline__ void sphere_light_radiance2(const SphereEmitter &E, V ldir, float d2, f2 ru, f2 rv, f2 &rad_x, f2 &rad_y, f2 &rad_z) {
    const f2 phi = (2.0f * 3.14159265359f) * ru;
    const f2 cos_t = 1.0f - 2.0f * rv;
    const f2 s2 = 1.0f - cos_t * cos_t;
    const f2 sin_t = f2{rt_sqrt(max_std(0.0f, s2.x)), rt_sqrt(max_std(0.0f, s2.y))};
    f2 sn, cs;
    rt_sincos2(phi, sn, cs);
    const f2 ux = sin_t * cs, uy = sin_t * sn, uz = cos_t;
    const f2 vx = (E.n[0] * ux + E.n[1] * uy) + E.n[2] * uz;
    const f2 vy = (E.n[3] * ux + E.n[4] * uy) + E.n[5] * uz;
    const f2 vz = (E.n[6] * ux + E.n[7] * uy) + E.n[8] * uz;
    const f2 len2 = (vx * vx + vy * vy) + vz * vz;
    // normalize(): the zero vector when |v|^2 <= 1e-12, else v * (1 / sqrt(|v|^2))
    const bool ok0 = len2.x > 1e-12f, ok1 = len2.y > 1e-12f;
    const f2 inv = f2{ok0 ? rt_div(1.0f, rt_sqrt(len2.x)) : 0.0f, ok1 ? rt_div(1.0f, rt_sqrt(len2.y)) : 0.0f};
    f2 nx = vx * inv, ny = vy * inv, nz = vz * inv;
    if (!ok0) nx.x = 0.0f, ny.x = 0.0f, nz.x = 0.0f;
    if (!ok1) nx.y = 0.0f, ny.y = 0.0f, nz.y = 0.0f;
    const V toward = scl(ldir, -1.0f);
    const f2 cos_l = (nx * toward.x + ny * toward.y) + nz * toward.z;
    const f2 num = E.area * cos_l;
    const f2 g = f2{cos_l.x <= 0.0f ? 0.0f : rt_div(num.x, d2), cos_l.y <= 0.0f ? 0.0f : rt_div(num.y, d2)};
    rad_x = E.color[0] * g;
    rad_y = E.color[1] * g;
    rad_z = E.color[2] * g;
    // a sample that faces away returns exactly (0, 0, 0), as the early return of sphere_light_radiance does
    if (cos_l.x <= 0.0f) rad_x.x = 0.0f, rad_y.x = 0.0f, rad_z.x = 0.0f;
    if (cos_l.y <= 0.0f) rad_x.y = 0.0f, rad_y.y = 0.0f, rad_z.y = 0.0f;
}

// `light_sphere`: the sphere of a sphere light (S.spheres[L.sphere] or a staged copy of it); unused otherwise.
__device__ __forceinline__ void light_sample(const DevScene &S, const rt_light &L, const rt_sphere *light_sphere, V p, V normal, float ru,
                                             float rv, V &radiance, V &ldir, float &dist, Counters &c) {
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
            float d2;
            sphere_light_geometry(*light_sphere, p, ldir, dist, d2);
            radiance = sphere_light_radiance(L, *light_sphere, ldir, d2, ru, rv);
            return;
        }
    }
    ldir = mk(0, 0, 0);
    dist = 0.f;
}

// calculate_color, src/ray_trace.cpp:154-406 (the last-blocker cache :158-159,207-226 is not used).
// The five BRDFs of calculate_color (src/ray_trace.cpp:246-389): diffuse and specular factors for one light
// direction, before multiplication by the incident radiance.
// The material enters as its BRDF selector (brdf | normalized << 8 | kdfresnel << 9) and Phong exponent.
__device__ __forceinline__ int brdf_code(const rt_material &m) { return m.brdf | (m.normalized ? 1 << 8 : 0) | (m.kdfresnel ? 1 << 9 : 0); }
__device__ __forceinline__ void brdf_terms(int code, float p_exp, V normal, V ldir, V view, V kd, V ks, V &fd, V &fs) {
    const int brdf = code & 0xff;
    const bool normalized = (code >> 8) & 1, kdfresnel = (code >> 9) & 1;
    float ndotl = max_std(dot(normal, ldir), 0.0f);
    V hv = normalize(add(ldir, view));
    fd = mk(0, 0, 0);
    fs = mk(0, 0, 0);
    const float PI = 3.14159265359f;
    switch (brdf) {
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
            if (normalized) {
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
            if (normalized) {
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
            if (kdfresnel)
                fd = mul(scl(kd, ndotl), sub(one, F));
            else
                fd = scl(kd, 1.0f / PI * ndotl);
            break;
        }
    }
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

}  // namespace rt
