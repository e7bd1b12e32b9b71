// BVH construction on the device for scenes that arrive without a tree (rt_scene_desc::n_bvh_nodes == 0 with
// bvh_prim_bounds given).
//
// The reference builds its tree on one host thread by recursive SAH splits (src/bvh.cpp:41-344); an animation rebuilds it
// for every frame.  This builder trades tree quality for build time: primitives are ordered along a 63-bit Morton curve of
// their box centres, the hierarchy comes from the common prefixes of neighbouring codes (one thread per interior node, no
// dependencies between them), boxes are fitted bottom-up, and the result is written straight into the records the trace
// kernel walks -- two child boxes per interior node, one leaf per primitive.  The tree differs from the reference's, so
// rays meet primitives in another order: the closest hit is the same, and only hits at exactly equal distance may resolve
// to another primitive (the reference keeps the first it finds, src/bvh.cpp:445).
//
// A node's children are ordered by the bit in which their codes first differ: the left child has 0 there, i.e. the lower
// coordinate on that bit's axis, which is what the walk's "near child first by the sign of the direction on the split axis"
// expects (src/bvh.cpp:459-474).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstring>
#include <vector>

#include "rt_bvh_gpu.h"
#include "rt_device.h"

namespace rt {

namespace {

#define BVH_HIP_OK(expr)                                                      \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);          \
            return 10;                                                        \
        }                                                                     \
    } while (0)

struct Scratch {   // device buffers of one build, released on every return path
    std::vector<void *> owned;
    ~Scratch() {
        for (void *p : owned) (void)hipFree(p);
    }
    template <class T> T *get(size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        owned.push_back(p);
        return (T *)p;
    }
    void release(void *p) {   // hands a buffer over to the caller
        for (auto &q : owned)
            if (q == p) q = nullptr;
    }
};

__device__ __forceinline__ unsigned long long spread21(unsigned v) {   // bit i of v to bit 3 i
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// code of primitive i: 21 bits per axis of its box centre within the box of all centres; x in bits 2, 5, ..., y 1, 4, ..., z 0, 3, ...
__global__ void __launch_bounds__(256) morton_kernel(const float *__restrict__ bounds, int n, float3 cmin, float3 inv_extent, unsigned long long *__restrict__ keys,
                                                      unsigned *__restrict__ order) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *b = bounds + 6 * (size_t)i;
    const float c[3] = {0.5f * (b[0] + b[3]), 0.5f * (b[1] + b[4]), 0.5f * (b[2] + b[5])};
    const float lo[3] = {cmin.x, cmin.y, cmin.z}, sc[3] = {inv_extent.x, inv_extent.y, inv_extent.z};
    unsigned q[3];
    for (int a = 0; a < 3; ++a) {
        float t = (c[a] - lo[a]) * sc[a];                 // 0 .. 1 (0 when the centres do not spread on this axis)
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        q[a] = min(2097151u, (unsigned)(t * 2097152.0f));
    }
    keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
    order[i] = (unsigned)i;
}

// length of the common prefix of the codes at sorted positions i and j; equal codes are told apart by their position
__device__ __forceinline__ int prefix(const unsigned long long *__restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// One thread per interior node (n - 1 of them): its range of sorted primitives, its split, its children.  child < 0:
// leaf ~child.  `axis_bit` = 1 << axis of the code bit in which the two children first differ.
__global__ void __launch_bounds__(256) hierarchy_kernel(const unsigned long long *__restrict__ keys, int n, int *__restrict__ left, int *__restrict__ right,
                                                         int *__restrict__ parent_of_node, int *__restrict__ parent_of_leaf, int *__restrict__ axis_bit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int floor_prefix = prefix(keys, n, i, i - d);
    int reach = 2;
    while (prefix(keys, n, i, i + reach * d) > floor_prefix) reach <<= 1;
    int len = 0;
    for (int t = reach >> 1; t > 0; t >>= 1)
        if (prefix(keys, n, i, i + (len + t) * d) > floor_prefix) len += t;
    const int j = i + len * d;
    const int node_prefix = prefix(keys, n, i, j);
    int s = 0;
    for (int t = (len + 1) >> 1;; t = (t + 1) >> 1) {
        if (prefix(keys, n, i, i + (s + t) * d) > node_prefix) s += t;
        if (t == 1) break;
    }
    const int split = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int l = lo == split ? ~split : split, r = hi == split + 1 ? ~(split + 1) : split + 1;
    left[i] = l, right[i] = r;
    if (l < 0) parent_of_leaf[~l] = i; else parent_of_node[l] = i;
    if (r < 0) parent_of_leaf[~r] = i; else parent_of_node[r] = i;
    if (i == 0) parent_of_node[0] = -1;
    // the first differing bit of the two halves: bit 63 - node_prefix of the code (x: bits 2 mod 3, y: 1, z: 0)
    axis_bit[i] = node_prefix < 64 ? 1 << (2 - (63 - node_prefix) % 3) : 1;
}

// One thread per leaf: climbs while it is the second to arrive, fitting each node's box from its children's.
__global__ void __launch_bounds__(256) fit_kernel(const float *__restrict__ bounds, const unsigned *__restrict__ order, int n, const int *__restrict__ left,
                                                   const int *__restrict__ right, const int *__restrict__ parent_of_node, const int *__restrict__ parent_of_leaf,
                                                   float *__restrict__ node_box, unsigned *__restrict__ arrived, int *__restrict__ deepest) {
    const int leaf = blockIdx.x * 256 + threadIdx.x;
    if (leaf >= n) return;
    int node = parent_of_leaf[leaf], steps = 1;
    for (int up = node; up >= 0 && parent_of_node[up] >= 0; up = parent_of_node[up]) ++steps;
    atomicMax(deepest, steps);
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;   // the sibling's subtree is not finished: that thread goes on
        __threadfence();
        float box[6] = {3.4e38f, 3.4e38f, 3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
        const int kids[2] = {left[node], right[node]};
        for (int k = 0; k < 2; ++k) {
            const volatile float *b = kids[k] < 0 ? bounds + 6 * (size_t)order[~kids[k]] : node_box + 6 * (size_t)kids[k];
            for (int a = 0; a < 3; ++a) {
                box[a] = fminf(box[a], b[a]);
                box[3 + a] = fmaxf(box[3 + a], b[3 + a]);
            }
        }
        for (int a = 0; a < 6; ++a) node_box[6 * (size_t)node + a] = box[a];
        node = parent_of_node[node];
    }
}

// The record of interior node i (DevScene::nodes): both children's boxes, their references, the split axis.
__global__ void __launch_bounds__(256) records_kernel(const float *__restrict__ bounds, const unsigned *__restrict__ order, int n, const int *__restrict__ left,
                                                       const int *__restrict__ right, const int *__restrict__ axis_bit, const float *__restrict__ node_box,
                                                       float4 *__restrict__ nodes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int kids[2] = {left[i], right[i]};
    float box[2][6];
    int ref[2];
    for (int k = 0; k < 2; ++k) {
        const float *b = kids[k] < 0 ? bounds + 6 * (size_t)order[~kids[k]] : node_box + 6 * (size_t)kids[k];
        for (int a = 0; a < 6; ++a) box[k][a] = b[a];
        ref[k] = kids[k] < 0 ? (int)(0x80000000u | (unsigned)~kids[k]) : kids[k];
    }
    float4 *o = nodes + 4 * (size_t)i;
    o[0] = make_float4(box[0][0], box[0][1], box[0][2], box[0][3]);
    o[1] = make_float4(box[0][4], box[0][5], box[1][0], box[1][1]);
    o[2] = make_float4(box[1][2], box[1][3], box[1][4], box[1][5]);
    o[3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), __int_as_float(axis_bit[i]), 0.f);
}

// Leaf entry and shading record of sorted position j (what rt_scene_create derives on the host for a tree that comes with
// the scene: scaled vertex and edges as the reference computes them, src/shapes.cpp:22-33,236-237; the face normal).
__global__ void __launch_bounds__(256) entries_kernel(const rt_bvh_prim *__restrict__ in, const float *__restrict__ bounds, const unsigned *__restrict__ order, int n,
                                                       const rt_triangle *__restrict__ triangles, int n_triangles, const rt_instance *__restrict__ instances,
                                                       float id_scale, float4 *__restrict__ prims, int4 *__restrict__ prim_meta) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const unsigned src = order[j];
    const rt_bvh_prim p = in[src];
    const float *b = bounds + 6 * (size_t)src;
    float4 *e = prims + PRIM_STRIDE * (size_t)j;
    int word;
    if (p.shape < n_triangles) {
        const rt_triangle &t = triangles[p.shape];
        float c1s[3], e1[3], e2[3];
        for (int a = 0; a < 3; ++a) {
            c1s[a] = t.c1[a] * 1000.0f;
            e1[a] = t.c2[a] * 1000.0f - c1s[a];
            e2[a] = t.c3[a] * 1000.0f - c1s[a];
        }
        const rt_instance &in_ = instances[p.instance];
        const bool plain = plain_placement(in_, id_scale);
        word = p.instance | (plain ? 0 : PRIM_XFORM_BIT);
        e[0] = make_float4(c1s[0], c1s[1], c1s[2], e1[0]);
        e[1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
        e[2] = make_float4(e2[2], __int_as_float(word | PRIM_LAST_BIT), b[0], b[1]);
        const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
        const float len2 = (nx * nx + ny * ny) + nz * nz;
        const bool ok = len2 > 1e-12f;
        const float inv = ok ? rt_div(1.0f, rt_sqrt(len2)) : 0.0f;
        prim_meta[2 * (size_t)j] = make_int4(word, p.shape, in_.ref_id, t.smooth ? 1 : 0);
        prim_meta[2 * (size_t)j + 1] = make_int4(__float_as_int(ok ? nx * inv : 0.0f), __float_as_int(ok ? ny * inv : 0.0f), __float_as_int(ok ? nz * inv : 0.0f), 0);
    } else {
        const int sphere = p.shape - n_triangles;
        e[0] = make_float4(0, 0, 0, 0);
        e[1] = make_float4(0, 0, 0, 0);
        e[2] = make_float4(0, __int_as_float((int)(0x80000000u | (unsigned)sphere | (unsigned)PRIM_LAST_BIT)), b[0], b[1]);
        prim_meta[2 * (size_t)j] = make_int4(-1 - sphere, p.shape, 123456, 0);
        prim_meta[2 * (size_t)j + 1] = make_int4(0, 0, 0, 0);
    }
    e[3] = make_float4(b[2], b[3], b[4], b[5]);
}

}  // namespace

int build_bvh_on_device(const rt_bvh_prim *prims, const float *bounds, int n, const rt_triangle *d_triangles, int n_triangles, const rt_instance *d_instances,
                        float id_scale, int max_depth, DeviceBvh &out, std::string &err) {
    out = DeviceBvh();
    if (n <= 0) return 0;
    // box of the scene and of the primitives' centres (one pass on the host, while the arrays are at hand)
    float smin[3] = {3.4e38f, 3.4e38f, 3.4e38f}, smax[3] = {-3.4e38f, -3.4e38f, -3.4e38f}, cmin[3], cmax[3];
    memcpy(cmin, smin, sizeof cmin);
    memcpy(cmax, smax, sizeof cmax);
    for (int i = 0; i < n; ++i) {
        const float *b = bounds + 6 * (size_t)i;
        for (int a = 0; a < 3; ++a) {
            if (!(b[a] <= b[3 + a]) || !std::isfinite(b[a]) || !std::isfinite(b[3 + a])) {
                err = "rt_scene_create: a primitive's box is not finite";
                return 3;
            }
            smin[a] = std::min(smin[a], b[a]);
            smax[a] = std::max(smax[a], b[3 + a]);
            const float c = 0.5f * (b[a] + b[3 + a]);
            cmin[a] = std::min(cmin[a], c);
            cmax[a] = std::max(cmax[a], c);
        }
    }
    memcpy(out.root_min, smin, sizeof smin);
    memcpy(out.root_max, smax, sizeof smax);
    Scratch mem;
    rt_bvh_prim *d_in = mem.get<rt_bvh_prim>((size_t)n);
    float *d_bounds = mem.get<float>(6 * (size_t)n);
    float4 *d_prims = mem.get<float4>(PRIM_STRIDE * (size_t)n);
    int4 *d_meta = mem.get<int4>(2 * (size_t)n);
    unsigned *order_in = mem.get<unsigned>((size_t)n), *order = mem.get<unsigned>((size_t)n);
    unsigned long long *keys_in = mem.get<unsigned long long>((size_t)n), *keys = mem.get<unsigned long long>((size_t)n);
    if (!d_in || !d_bounds || !d_prims || !d_meta || !order_in || !order || !keys_in || !keys) {
        err = "rt_scene_create: out of device memory for the tree build";
        return 10;
    }
    BVH_HIP_OK(hipMemcpy(d_in, prims, (size_t)n * sizeof(rt_bvh_prim), hipMemcpyHostToDevice));
    BVH_HIP_OK(hipMemcpy(d_bounds, bounds, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    const unsigned blocks_n = (unsigned)((n + 255) / 256);
    const float3 lo = make_float3(cmin[0], cmin[1], cmin[2]);
    const float3 inv = make_float3(cmax[0] > cmin[0] ? 1.0f / (cmax[0] - cmin[0]) : 0.0f, cmax[1] > cmin[1] ? 1.0f / (cmax[1] - cmin[1]) : 0.0f,
                                   cmax[2] > cmin[2] ? 1.0f / (cmax[2] - cmin[2]) : 0.0f);
    hipLaunchKernelGGL(morton_kernel, dim3(blocks_n), dim3(256), 0, 0, d_bounds, n, lo, inv, keys_in, order_in);
    BVH_HIP_OK(hipGetLastError());
    {   // order the primitives along the curve
        size_t temp_bytes = 0;
        BVH_HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in, keys, order_in, order, n, 0, 63, 0));
        void *temp = mem.get<char>(temp_bytes);
        if (!temp) {
            err = "rt_scene_create: out of device memory for the tree build";
            return 10;
        }
        BVH_HIP_OK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys, order_in, order, n, 0, 63, 0));
    }
    hipLaunchKernelGGL(entries_kernel, dim3(blocks_n), dim3(256), 0, 0, d_in, d_bounds, order, n, d_triangles, n_triangles, d_instances, id_scale, d_prims, d_meta);
    BVH_HIP_OK(hipGetLastError());
    if (n == 1) {   // the root is the one leaf
        out.root_ref = (int)0x80000000u;
    } else {
        const int ni = n - 1;
        int *left = mem.get<int>((size_t)ni), *right = mem.get<int>((size_t)ni), *parent_of_node = mem.get<int>((size_t)ni), *parent_of_leaf = mem.get<int>((size_t)n),
            *axis_bit = mem.get<int>((size_t)ni), *deepest = mem.get<int>(1);
        float *node_box = mem.get<float>(6 * (size_t)ni);
        unsigned *arrived = mem.get<unsigned>((size_t)ni);
        float4 *d_nodes = mem.get<float4>(4 * (size_t)ni);
        if (!left || !right || !parent_of_node || !parent_of_leaf || !axis_bit || !deepest || !node_box || !arrived || !d_nodes) {
            err = "rt_scene_create: out of device memory for the tree build";
            return 10;
        }
        BVH_HIP_OK(hipMemsetAsync(arrived, 0, (size_t)ni * sizeof(unsigned), 0));
        BVH_HIP_OK(hipMemsetAsync(deepest, 0, sizeof(int), 0));
        const unsigned blocks_i = (unsigned)((ni + 255) / 256);
        hipLaunchKernelGGL(hierarchy_kernel, dim3(blocks_i), dim3(256), 0, 0, keys, n, left, right, parent_of_node, parent_of_leaf, axis_bit);
        hipLaunchKernelGGL(fit_kernel, dim3(blocks_n), dim3(256), 0, 0, d_bounds, order, n, left, right, parent_of_node, parent_of_leaf, node_box, arrived, deepest);
        hipLaunchKernelGGL(records_kernel, dim3(blocks_i), dim3(256), 0, 0, d_bounds, order, n, left, right, axis_bit, node_box, d_nodes);
        BVH_HIP_OK(hipGetLastError());
        BVH_HIP_OK(hipMemcpy(&out.depth, deepest, sizeof(int), hipMemcpyDeviceToHost));
        if (out.depth > max_depth) {
            err = "rt_scene_create: the tree built on the device is " + std::to_string(out.depth) + " levels deep, more than the walk's stack of " +
                  std::to_string(max_depth) + " holds (many primitives with the same centre); build this scene's tree on the host";
            return 7;
        }
        out.nodes = d_nodes;
        out.n_interior = ni;
        out.root_ref = 0;
        mem.release(d_nodes);
    }
    BVH_HIP_OK(hipDeviceSynchronize());
    out.prims = d_prims;
    out.prim_meta = d_meta;
    mem.release(d_prims);
    mem.release(d_meta);
    return 0;
}

}  // namespace rt
