// BVH construction on the device for scenes that arrive without a tree (rt_scene_desc::n_bvh_nodes == 0 with
// bvh_prim_bounds given).
//
// The reference builds its tree on one host thread by recursive SAH splits (src/bvh.cpp:41-344); an animation rebuilds it
// for every frame.  Here the primitives are ordered along a 63-bit Morton curve of their box centres and the tree is grown
// bottom-up by clustering along that order (parallel locally-ordered clustering): every cluster looks RADIUS places to
// either side for the neighbour whose union with it has the least surface area, pairs that choose each other merge into a
// node, and the survivors are compacted in order for the next round -- some forty rounds for a million primitives, each a
// few short launches.  The surface area of the union is what the reference's split cost weighs (src/bvh.cpp:249-259), so the
// clusters that end up together are the ones a SAH split would keep together.  While the tree grows every node learns the
// cost of its subtree by the reference's rule (src/bvh.cpp:259-274: 0.125 + the children's costs weighted by their share of
// the node's area, against one unit per primitive for a leaf of at most four), and a node that is cheaper as a leaf becomes
// one: its primitives' entries are laid out next to each other and tested one after the other, like a leaf of the
// reference's tree.  The result is written straight into the records the trace kernel walks -- two child boxes per interior
// node, leaves of one to four primitives.  The tree differs from the reference's, so rays meet primitives in another order:
// the closest hit is the same, and only hits at exactly equal distance may resolve to another primitive (the reference
// keeps the first it finds, src/bvh.cpp:445).
//
// A node's children are ordered along the axis on which their centres lie furthest apart: the left child is the lower one,
// which is what the walk's "near child first by the sign of the direction on the split axis" expects (src/bvh.cpp:459-474).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstring>
#include <vector>

#include "rt_bvh_gpu.h"
#include "rt_device.h"
#include "rt_devmem.h"

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

// RT_CREATE_TIMING: stage times of a build on stderr (diagnostic; every mark waits for the device)
struct StageClock {
    const bool on = getenv("RT_CREATE_TIMING") != nullptr;
    std::chrono::high_resolution_clock::time_point last = std::chrono::high_resolution_clock::now();
    void mark(const char *what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::high_resolution_clock::now();
        fprintf(stderr, "[create] tree build: %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

struct Scratch {   // device buffers of one build (from the device's block cache), released on every return path
    std::vector<void *> owned;
    ~Scratch() {
        StageClock clock;
        (void)hipDeviceSynchronize();   // (a build that gave up may still have launches in flight; blocks go back only when idle)
        for (void *p : owned) device_blocks().free(p);
        clock.mark("buffers released");
    }
    template <class T> T *get(size_t n) {
        void *p = device_blocks().alloc(std::max<size_t>(n, 1) * sizeof(T));
        if (!p) return nullptr;
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

// ---- clustering along the Morton order -------------------------------------------------------------------------------
// A cluster: a node reference (leaf ~j of sorted position j, or interior node id), its box, and what is known of its
// subtree: {depth in interior nodes, primitives, cost (float bits), 1 when the subtree is one leaf}.
__device__ __forceinline__ float half_area(const float *b) {
    const float x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2];
    return x * y + x * z + y * z;
}
__device__ __forceinline__ float union_area(const float *a, const float *b) {
    const float x = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), y = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), z = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return x * y + x * z + y * z;
}

__global__ void __launch_bounds__(256) ploc_leaves_kernel(const float *__restrict__ bounds, const unsigned *__restrict__ order, int n, int *__restrict__ ref,
                                                           float *__restrict__ box, int4 *__restrict__ info) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    ref[j] = ~j;
    info[j] = make_int4(0, 1, __float_as_int(1.0f), 1);
    const float *b = bounds + 6 * (size_t)order[j];
    for (int a = 0; a < 6; ++a) box[6 * (size_t)j + a] = b[a];
}

// nearest[i]: the cluster within `radius` places of i whose union with i has the least area.  Ties go to the pair (2 m,
// 2 m + 1) first, then to the lower (smaller index, larger index): a total order on pairs, so the best pair of a round
// always choose each other and every round merges at least once -- and a pile of identical boxes pairs off in every round
// (a balanced tree) instead of growing a chain from one end.
// (The clusters' count of a round, state[0], and the nodes made so far, state[1], live on the device: the host reads them
// every few rounds only, and launches the rounds in between for the count it last saw -- lanes beyond the true count leave.)
__global__ void __launch_bounds__(256) ploc_nearest_kernel(const float *__restrict__ box, const int *__restrict__ state, int radius, int *__restrict__ nearest) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nc = state[0];
    if (i >= nc) return;
    float mine[6];
    for (int a = 0; a < 6; ++a) mine[a] = box[6 * (size_t)i + a];
    float best = 3.4e38f;
    int arg = -1;
    const int lo = max(0, i - radius), hi = min(nc - 1, i + radius);
    for (int k = lo; k <= hi; ++k) {
        if (k == i) continue;
        const float area = union_area(mine, box + 6 * (size_t)k);
        bool better = arg < 0 || area < best;
        if (!better && area == best) {
            const bool buddy_k = (i ^ k) == 1, buddy_arg = (i ^ arg) == 1;
            better = buddy_k != buddy_arg ? buddy_k : (min(i, k) < min(i, arg) || (min(i, k) == min(i, arg) && max(i, k) < max(i, arg)));
        }
        if (better) best = area, arg = k;
    }
    nearest[i] = arg;
}

// flags[i]: low word 1 when cluster i survives the round (alone, or as the lower partner of a merge), high word 1 when it
// is the lower partner of a merge; an exclusive scan of the flags gives the survivor's new place and the new node's number
__global__ void __launch_bounds__(256) ploc_flag_kernel(const int *__restrict__ nearest, const int *__restrict__ state, int bound, unsigned long long *__restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > bound) return;
    const int nc = state[0];
    if (i >= nc) {   // (the scan runs over bound + 1 flags: its last element holds the totals)
        flags[i] = 0ull;
        return;
    }
    const int j = nearest[i];
    const bool mutual = j >= 0 && nearest[j] == i;
    flags[i] = (mutual && i > j ? 0ull : 1ull) | (mutual && i < j ? 1ull << 32 : 0ull);
}

struct TreeArrays {   // per interior node id (in the order the nodes are made: children before parents)
    int *left, *right, *parent, *axis_bit;
    int *live;        // interior nodes of the subtree that the walk can reach (0 for a subtree that became a leaf)
    int4 *info;       // as a cluster's
    float *box;
    int *parent_of_leaf;   // per sorted position
    int4 *wide;       // for the four-box records: {reachable interior nodes of the subtree an even number of steps below this
                      // node (itself included), an odd number of steps below, stack entries a walk needs below this node
                      // were it the head of a record, unused}
};

__global__ void __launch_bounds__(256) ploc_merge_kernel(const int *__restrict__ nearest, const unsigned long long *__restrict__ flags,
                                                          const unsigned long long *__restrict__ scanned, const int *__restrict__ state, const int *__restrict__ ref_in,
                                                          const float *__restrict__ box_in, const int4 *__restrict__ info_in, int *__restrict__ ref_out,
                                                          float *__restrict__ box_out, int4 *__restrict__ info_out, TreeArrays T) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nc = state[0], first_node = state[1];
    if (i >= nc) return;
    const unsigned long long f = flags[i];
    if (!(f & 1ull)) return;   // merged into its partner
    const int place = (int)(scanned[i] & 0xffffffffull);
    if (!(f >> 32)) {          // survives as it is
        ref_out[place] = ref_in[i];
        info_out[place] = info_in[i];
        for (int a = 0; a < 6; ++a) box_out[6 * (size_t)place + a] = box_in[6 * (size_t)i + a];
        return;
    }
    const int j = nearest[i];
    const int id = first_node + (int)(scanned[i] >> 32);
    const float *bi = box_in + 6 * (size_t)i, *bj = box_in + 6 * (size_t)j;
    // the axis on which the two centres lie furthest apart; the lower one goes left
    int axis = 0;
    float far = -1.0f, diff = 0.0f;
    for (int a = 0; a < 3; ++a) {
        const float d = (bi[a] + bi[3 + a]) - (bj[a] + bj[3 + a]);
        if (fabsf(d) > far) far = fabsf(d), axis = a, diff = d;
    }
    const bool i_first = diff <= 0.0f;
    const int l = i_first ? ref_in[i] : ref_in[j], r = i_first ? ref_in[j] : ref_in[i];
    T.left[id] = l, T.right[id] = r;
    if (l < 0) T.parent_of_leaf[~l] = id; else T.parent[l] = id;
    if (r < 0) T.parent_of_leaf[~r] = id; else T.parent[r] = id;
    T.parent[id] = -1;   // (until a later round says otherwise)
    T.axis_bit[id] = 1 << axis;
    float u[6];
    for (int a = 0; a < 3; ++a) u[a] = fminf(bi[a], bj[a]), u[3 + a] = fmaxf(bi[3 + a], bj[3 + a]);
    for (int a = 0; a < 6; ++a) T.box[6 * (size_t)id + a] = u[a], box_out[6 * (size_t)place + a] = u[a];
    // cost of the subtree by the reference's rule (src/bvh.cpp:259-274), and whether it is cheaper as one leaf
    const int4 ii = info_in[i], ij = info_in[j];
    const int count = ii.y + ij.y;
    const float area = half_area(u);
    float cost = 0.125f + (area > 0.0f ? (half_area(bi) * __int_as_float(ii.z) + half_area(bj) * __int_as_float(ij.z)) / area : (float)count);
    const bool leaf = count <= 4 && ii.w && ij.w && (float)count <= cost;
    if (leaf) cost = (float)count;
    const int4 mine = make_int4(leaf ? 0 : 1 + max(ii.x, ij.x), count, __float_as_int(cost), leaf ? 1 : 0);
    T.info[id] = mine;
    T.live[id] = leaf ? 0 : 1 + (l < 0 ? 0 : T.live[l]) + (r < 0 ? 0 : T.live[r]);
    {   // (children and grandchildren were made in earlier rounds)
        int4 w = make_int4(0, 0, 0, 0);
        if (!leaf) {
            const int4 wl = l < 0 ? make_int4(0, 0, 0, 0) : T.wide[l], wr = r < 0 ? make_int4(0, 0, 0, 0) : T.wide[r];
            w.x = 1 + wl.y + wr.y;
            w.y = wl.x + wr.x;
            // a record headed here holds a leaf child itself and an interior child's two children: the walk keeps all but one
            // of them waiting while it goes down the deepest (rt_wide.h, build_wide_nodes)
            int present = 0, deepest = 0;
            const int kids[2] = {l, r};
            for (int k = 0; k < 2; ++k) {
                const int c = kids[k];
                if (c < 0 || T.info[c].w) {
                    present += 1;
                    continue;
                }
                present += 2;
                const int grand[2] = {T.left[c], T.right[c]};
                for (int g = 0; g < 2; ++g)
                    if (grand[g] >= 0 && !T.info[grand[g]].w) deepest = max(deepest, T.wide[grand[g]].z);
            }
            w.z = present - 1 + deepest;
        }
        T.wide[id] = w;
    }
    ref_out[place] = id;
    info_out[place] = mine;
}

// After a round: the survivors are the next round's clusters, the merges the nodes made.
__global__ void ploc_advance_kernel(const unsigned long long *__restrict__ scanned, int bound, int *__restrict__ state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long totals = scanned[bound];
    state[0] = (int)(totals & 0xffffffffull);
    state[1] += (int)(totals >> 32);
}

// Where every primitive's entry goes: the tree's leaves from left to right, a leaf's primitives next to each other.  One
// thread per sorted position climbs to the root, adding up the primitives of the subtrees to its left; the topmost
// ancestor that is a leaf (if any) is the leaf it belongs to.  place[j] = its entry, leaf_of[j] = that ancestor or -1,
// and for such an ancestor first_entry[] = the place of its first primitive (every one of its primitives writes the same value).
__global__ void __launch_bounds__(256) ploc_places_kernel(int n, TreeArrays T, int *__restrict__ place, int *__restrict__ leaf_of, int *__restrict__ first_entry) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int before = 0, child = ~j, leaf = -1, before_in_leaf = 0;
    for (int node = T.parent_of_leaf[j]; node >= 0; node = T.parent[node]) {
        if (T.right[node] == child) {
            const int l = T.left[node];
            before += l < 0 ? 1 : T.info[l].y;
        }
        if (T.info[node].w) leaf = node, before_in_leaf = before;
        child = node;
    }
    place[j] = before;
    leaf_of[j] = leaf;
    if (leaf >= 0) first_entry[leaf] = before - before_in_leaf;
}

// Where a node's record goes.  The walk reads a child right after its parent, so the records stand in the order of the
// reference's array (src/bvh.cpp:316-344: a node, its left subtree, its right subtree): a left child lies next to its
// parent.  The top of the tree -- the nodes less than TOP_LEVELS steps below the root, which every ray visits and the trace
// kernel keeps in LDS -- comes first, level by level (heap numbering: 2 h + 1, 2 h + 2).  One thread per node climbs to the
// root to find both numbers.  Nodes inside a subtree that became a leaf get no record.
constexpr int TOP_LEVELS = 7;                      // 127 records
constexpr int TOP_RECORDS = (1 << TOP_LEVELS) - 1;
// The first records of the four-box tree, for the trace kernel's LDS copy: the heads a breadth-first walk from the root meets
// first (the order build_wide_nodes of rt_wide.h gives the host's tree).  One thread: at most TOP_RECORDS short steps.
// wide_top[] is -1 everywhere when the kernel starts; queue[] holds 4 * TOP_RECORDS + 4 entries.
__device__ __forceinline__ bool one_leaf(const TreeArrays &T, int c) { return c < 0 || T.info[c].w != 0; }
__global__ void wide_top_kernel(int root, TreeArrays T, int *__restrict__ wide_top, int *__restrict__ queue, int *__restrict__ n_top_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int size = 0, count = 0;
    queue[size++] = root;
    for (int at = 0; at < size && count < TOP_RECORDS; ++at) {
        const int i = queue[at];
        wide_top[i] = count++;
        const int kids[2] = {T.left[i], T.right[i]};
        for (int k = 0; k < 2; ++k) {
            if (one_leaf(T, kids[k])) continue;
            const int grand[2] = {T.left[kids[k]], T.right[kids[k]]};
            for (int g = 0; g < 2; ++g)
                if (!one_leaf(T, grand[g])) queue[size++] = grand[g];
        }
    }
    *n_top_out = count;
}

__global__ void __launch_bounds__(256) ploc_record_places_kernel(int ni, TreeArrays T, const int *__restrict__ wide_top, int *__restrict__ record_of,
                                                                  int *__restrict__ wide_record_of) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= ni) return;
    int before = 0, depth = 0, child = id;
    int heads_before = 0;     // record heads before this node in the same order, were this node a head (an even number of steps below the root)
    unsigned path = 0;        // the turns from the root down, the last one in bit 0
    bool reachable = !T.info[id].w;
    for (int node = T.parent[id]; node >= 0; node = T.parent[node]) {
        const bool is_right = T.right[node] == child;
        const bool even_step = (depth & 1) != 0;   // this ancestor is depth + 1 steps up
        if (is_right) {
            const int l = T.left[node];
            if (l >= 0) {
                before += T.live[l];
                const int4 w = T.wide[l];            // (the sibling is one step nearer than the ancestor)
                heads_before += even_step ? w.y : w.x;
            }
        }
        before += 1;
        heads_before += even_step ? 1 : 0;
        if (depth < 32) path |= (is_right ? 1u : 0u) << depth;
        ++depth;
        if (T.info[node].w) reachable = false;
        child = node;
    }
    if (wide_record_of) wide_record_of[id] = !reachable || (depth & 1) ? -1 : (wide_top[id] >= 0 ? wide_top[id] : TOP_RECORDS + heads_before);
    if (!reachable) {
        record_of[id] = -1;
        return;
    }
    if (depth < TOP_LEVELS) {
        int h = 0;
        for (int k = depth - 1; k >= 0; --k) h = 2 * h + 1 + (int)((path >> k) & 1u);
        record_of[id] = h;
    } else {
        record_of[id] = TOP_RECORDS + before;
    }
}

__global__ void __launch_bounds__(256) ploc_records_kernel(const float *__restrict__ bounds, const unsigned *__restrict__ order, int ni, TreeArrays T,
                                                            const int *__restrict__ place, const int *__restrict__ first_entry, const int *__restrict__ record_of,
                                                            float4 *__restrict__ nodes) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= ni || record_of[id] < 0) return;
    const int kids[2] = {T.left[id], T.right[id]};
    float box[2][6];
    int ref[2];
    for (int k = 0; k < 2; ++k) {
        const float *b = kids[k] < 0 ? bounds + 6 * (size_t)order[~kids[k]] : T.box + 6 * (size_t)kids[k];
        for (int a = 0; a < 6; ++a) box[k][a] = b[a];
        if (kids[k] < 0)
            ref[k] = (int)(0x80000000u | (unsigned)place[~kids[k]]);
        else if (T.info[kids[k]].w)
            ref[k] = (int)(0x80000000u | (unsigned)first_entry[kids[k]]);
        else
            ref[k] = record_of[kids[k]];
    }
    float4 *o = nodes + 4 * (size_t)record_of[id];
    o[0] = make_float4(box[0][0], box[0][1], box[0][2], box[0][3]);
    o[1] = make_float4(box[0][4], box[0][5], box[1][0], box[1][1]);
    o[2] = make_float4(box[1][2], box[1][3], box[1][4], box[1][5]);
    o[3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), __int_as_float(T.axis_bit[id]), 0.f);
}

// The four-box record headed by node `id` (DevScene::wnodes; the host's fill_wide_record in rt_wide.h makes the same record
// from the same boxes): a child that is a leaf fills one slot, an interior child gives its own two children, the boxes are
// bytes on a grid of power-of-two steps from the lowest corner, rounded outwards -- the decoded coordinate is
// fmaf(q, step, origin) here and in the trace kernel, and is never inside the exact box.  *unsound is set when a record
// cannot be made (a grid step beyond the exponent range).
__device__ bool quantise_axis(const float lo_of[4], const float hi_of[4], const bool present[4], float &origin, float &step_out, unsigned &qlo, unsigned &qhi) {
    float lo = INFINITY, hi = -INFINITY;
    for (int k = 0; k < 4; ++k)
        if (present[k]) lo = fminf(lo, lo_of[k]), hi = fmaxf(hi, hi_of[k]);
    origin = lo;
    // the grid step is never so small that origin + 255 steps rounds back to the origin: an empty slot's inverted box
    // (255..0) then misses on every axis the ray moves along
    int e = max(-100, ilogbf(fmaxf(fmaxf(fabsf(lo), fabsf(hi)), 1e-30f)) - 22);
    const double extent = (double)hi - (double)lo;
    if (extent > 0) e = max(ilogb(extent / 255.0) + 1, e);   // 2^e > extent / 255
    for (;; ++e) {   // the smallest step with which every box fits 0..255
        if (e > 126) return false;
        const float step = ldexpf(1.0f, e);
        const double per_step = ldexp(1.0, -e);   // a power of two: the products below are exact
        unsigned lo_bytes = 0, hi_bytes = 0;
        bool fits = true;
        for (int k = 0; k < 4 && fits; ++k) {
            if (!present[k]) {
                lo_bytes |= 255u << (8 * k);
                continue;
            }
            int ql = (int)fmin(255.0, floor(((double)lo_of[k] - (double)lo) * per_step));
            while (ql > 0 && fmaf((float)ql, step, lo) > lo_of[k]) --ql;
            int qh = (int)fmin(256.0, ceil(((double)hi_of[k] - (double)lo) * per_step));
            while (qh <= 255 && fmaf((float)qh, step, lo) < hi_of[k]) ++qh;
            if (qh > 255) fits = false;
            lo_bytes |= (unsigned)ql << (8 * k);
            hi_bytes |= (unsigned)(qh & 255) << (8 * k);
        }
        if (fits) {
            qlo = lo_bytes, qhi = hi_bytes, step_out = step;
            return true;
        }
    }
}

// Quantises the (up to) four boxes of a record and writes it; false when the grid cannot hold them.
__device__ bool write_wide_record(const float lo_of[3][4], const float hi_of[3][4], const bool present[4], const unsigned ref[4], unsigned head_bits, float4 *o) {
    float origin[3], step[3];
    unsigned qlo[3], qhi[3];
    for (int a = 0; a < 3; ++a)
        if (!quantise_axis(lo_of[a], hi_of[a], present, origin[a], step[a], qlo[a], qhi[a])) return false;
    o[0] = make_float4(origin[0], origin[1], origin[2], __uint_as_float(__float_as_uint(step[0]) | head_bits));
    o[1] = make_float4(__uint_as_float(ref[0]), __uint_as_float(ref[1]), __uint_as_float(ref[2]), __uint_as_float(ref[3]));
    o[2] = make_float4(__uint_as_float(qlo[0]), __uint_as_float(qhi[0]), __uint_as_float(qlo[1]), __uint_as_float(qhi[1]));
    o[3] = make_float4(__uint_as_float(qlo[2]), __uint_as_float(qhi[2]), step[1], step[2]);
    return true;
}

__global__ void __launch_bounds__(256) ploc_wide_records_kernel(const float *__restrict__ bounds, const unsigned *__restrict__ order, int ni, TreeArrays T,
                                                                 const int *__restrict__ place, const int *__restrict__ first_entry,
                                                                 const int *__restrict__ wide_record_of, float4 *__restrict__ wnodes, int *__restrict__ unsound) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= ni || wide_record_of[id] < 0) return;
    const int kids[2] = {T.left[id], T.right[id]};
    int slot[4];                 // node or ~position; `present` says whether there is one
    bool present[4];
    unsigned head_bits = (unsigned)T.axis_bit[id];
    for (int k = 0; k < 2; ++k) {
        const int c = kids[k];
        if (one_leaf(T, c)) {
            slot[2 * k] = c, present[2 * k] = true;
            slot[2 * k + 1] = 0, present[2 * k + 1] = false;
        } else {
            slot[2 * k] = T.left[c], slot[2 * k + 1] = T.right[c];
            present[2 * k] = present[2 * k + 1] = true;
            head_bits |= (unsigned)T.axis_bit[c] << (3 + 3 * k);
        }
    }
    unsigned ref[4];
    float lo_of[3][4], hi_of[3][4];
    for (int k = 0; k < 4; ++k) {
        if (!present[k]) {
            ref[k] = 0x40000000u;   // the walk's "pop" reference, harmless if ever followed
            for (int a = 0; a < 3; ++a) lo_of[a][k] = hi_of[a][k] = 0.0f;
            continue;
        }
        const int c = slot[k];
        const float *b = c < 0 ? bounds + 6 * (size_t)order[~c] : T.box + 6 * (size_t)c;
        for (int a = 0; a < 3; ++a) lo_of[a][k] = b[a], hi_of[a][k] = b[3 + a];
        if (c < 0)
            ref[k] = 0x80000000u | (unsigned)place[~c];
        else if (T.info[c].w)
            ref[k] = 0x80000000u | (unsigned)first_entry[c];
        else
            ref[k] = (unsigned)wide_record_of[c];
    }
    if (!write_wide_record(lo_of, hi_of, present, ref, head_bits, wnodes + 4 * (size_t)wide_record_of[id])) *unsound = 1;
}

// Leaf entry and shading record of sorted position j, written to its place: scaled vertex and edges as the reference
// computes them (src/shapes.cpp:22-33,236-237), the face normal.  For a tree built here the entry carries the box of its leaf
// and says whether it is the leaf's last; for a tree that came with the scene (`order`, `bounds`, `place` null: entry j is
// primitive j) leaf_boxes_kernel adds both afterwards.
__global__ void __launch_bounds__(256) entries_kernel(const rt_bvh_prim *__restrict__ in, const float *__restrict__ bounds, const unsigned *__restrict__ order, int n,
                                                       const rt_triangle *__restrict__ triangles, int n_triangles, const rt_instance *__restrict__ instances,
                                                       float id_scale, const int *__restrict__ place, const int *__restrict__ leaf_of, const int *__restrict__ first_entry,
                                                       const int4 *__restrict__ node_info, const float *__restrict__ node_box, float4 *__restrict__ prims,
                                                       int4 *__restrict__ prim_meta) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const unsigned src = order ? order[j] : (unsigned)j;
    const rt_bvh_prim p = in[src];
    const int at = place ? place[j] : j, leaf = leaf_of ? leaf_of[j] : -1;
    const float no_box[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *b = leaf >= 0 ? node_box + 6 * (size_t)leaf : (bounds ? bounds + 6 * (size_t)src : no_box);
    const bool last = bounds && (leaf < 0 || at == first_entry[leaf] + node_info[leaf].y - 1);
    const int last_bit = last ? PRIM_LAST_BIT : 0;
    float4 *e = prims + PRIM_STRIDE * (size_t)at;
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
        e[2] = make_float4(e2[2], __int_as_float(word | last_bit), b[0], b[1]);
        const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
        const float len2 = (nx * nx + ny * ny) + nz * nz;
        const bool ok = len2 > 1e-12f;
        const float inv = ok ? rt_div(1.0f, rt_sqrt(len2)) : 0.0f;
        prim_meta[2 * (size_t)at] = make_int4(word, p.shape, in_.ref_id, t.smooth ? 1 : 0);
        prim_meta[2 * (size_t)at + 1] = make_int4(__float_as_int(ok ? nx * inv : 0.0f), __float_as_int(ok ? ny * inv : 0.0f), __float_as_int(ok ? nz * inv : 0.0f), 0);
    } else {
        const int sphere = p.shape - n_triangles;
        e[0] = make_float4(0, 0, 0, 0);
        e[1] = make_float4(0, 0, 0, 0);
        e[2] = make_float4(0, __int_as_float((int)(0x80000000u | (unsigned)sphere | (unsigned)last_bit)), b[0], b[1]);
        prim_meta[2 * (size_t)at] = make_int4(-1 - sphere, p.shape, 123456, 0);
        prim_meta[2 * (size_t)at + 1] = make_int4(0, 0, 0, 0);
    }
    e[3] = make_float4(b[2], b[3], b[4], b[5]);
}

// ---- records of a tree that came with the scene ------------------------------------------------------------------------
// Every entry of a leaf carries the leaf's box, the last one the end mark.
__global__ void __launch_bounds__(256) leaf_boxes_kernel(const rt_bvh_node *__restrict__ nodes, int n_nodes, float4 *__restrict__ prims) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const rt_bvh_node n = nodes[i];
    for (int k = 0; k < n.n_prims; ++k) {
        float4 *e = prims + PRIM_STRIDE * (size_t)(n.offset + k);
        float4 w = e[2];
        w.z = n.bmin[0], w.w = n.bmin[1];
        if (k == n.n_prims - 1) w.y = __int_as_float(__float_as_int(w.y) | PRIM_LAST_BIT);
        e[2] = w;
        e[3] = make_float4(n.bmin[2], n.bmax[0], n.bmax[1], n.bmax[2]);
    }
}

// The two-box record of interior node i at the place `record_of[i]` gives it: both children's boxes, their references (a
// record, or a leaf's first entry) and the split axis.
__global__ void __launch_bounds__(256) node_records_kernel(const rt_bvh_node *__restrict__ nodes, int n_nodes, const int *__restrict__ record_of, float4 *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const rt_bvh_node n = nodes[i];
    if (n.n_prims != 0) return;
    const rt_bvh_node L = nodes[i + 1], R = nodes[n.offset];
    const int lr = L.n_prims == 0 ? record_of[i + 1] : (int)(0x80000000u | (unsigned)L.offset);
    const int rr = R.n_prims == 0 ? record_of[n.offset] : (int)(0x80000000u | (unsigned)R.offset);
    float4 *o = out + 4 * (size_t)record_of[i];
    o[0] = make_float4(L.bmin[0], L.bmin[1], L.bmin[2], L.bmax[0]);
    o[1] = make_float4(L.bmax[1], L.bmax[2], R.bmin[0], R.bmin[1]);
    o[2] = make_float4(R.bmin[2], R.bmax[0], R.bmax[1], R.bmax[2]);
    o[3] = make_float4(__int_as_float(lr), __int_as_float(rr), __int_as_float(1 << n.axis), 0.f);
}

// ---- four-box records of a tree that came with the scene ---------------------------------------------------------------
// The reference's array holds a node, its left subtree, its right subtree (src/bvh.cpp:316-344): the left child of interior
// node i is i + 1, the right one nodes[i].offset.
__global__ void __launch_bounds__(256) tree_interior_kernel(const rt_bvh_node *__restrict__ nodes, int n_nodes, int *__restrict__ interior) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_nodes) interior[i] = nodes[i].n_prims == 0 ? 1 : 0;
}
__global__ void __launch_bounds__(256) tree_parents_kernel(const rt_bvh_node *__restrict__ nodes, int n_nodes, int *__restrict__ parent) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    if (i == 0) parent[0] = -1;
    if (nodes[i].n_prims != 0) return;
    parent[i + 1] = i;
    parent[nodes[i].offset] = i;
}

// head[i] = 1 when node i heads a record: an interior node an even number of steps below the root.  On its way to the root
// a head also adds up what a walk keeps waiting on the stack above it -- every record holds its slots but one -- and the
// largest such sum over all heads is the stack the deepest walk needs (build_wide_nodes of rt_wide.h computes the same
// number bottom-up).
__global__ void __launch_bounds__(256) tree_heads_kernel(const rt_bvh_node *__restrict__ nodes, int n_nodes, const int *__restrict__ parent, int *__restrict__ head,
                                                          int *__restrict__ stack_need) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    if (nodes[i].n_prims != 0) {
        head[i] = 0;
        return;
    }
    auto waiting = [&](int a) {   // slots of the record a heads, less the one the walk goes on with
        const int l = a + 1, r = nodes[a].offset;
        return 1 + (nodes[l].n_prims == 0 ? 1 : 0) + (nodes[r].n_prims == 0 ? 1 : 0);
    };
    int depth = 0, even = waiting(i), odd = 0;   // sums over the ancestors an even / odd number of steps up (this node: zero steps)
    for (int a = parent[i]; a >= 0; a = parent[a]) {
        ++depth;
        if (depth & 1) odd += waiting(a); else even += waiting(a);
    }
    const bool is_head = (depth & 1) == 0;
    head[i] = is_head ? 1 : 0;
    if (is_head) atomicMax(stack_need, even);
}

// where[i]: the record of node i when head[i] is set (a head of the four-box records, or any interior node for the two-box
// records) -- its number among such nodes in array order behind the places of the top, unless it is one of the top nodes
// (listed breadth-first in `top`), whose place is its number there.
__global__ void __launch_bounds__(256) tree_places_kernel(int n_nodes, const int *__restrict__ head, const int *__restrict__ rank, int top_places, int *__restrict__ where) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    where[i] = head[i] ? top_places + rank[i] : -1;
}
__global__ void tree_top_places_kernel(const int *__restrict__ top, int n_top, int *__restrict__ where) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_top) where[top[k]] = k;
}

__global__ void __launch_bounds__(256) tree_wide_records_kernel(const rt_bvh_node *__restrict__ nodes, int n_nodes, const int *__restrict__ where, float4 *__restrict__ wnodes,
                                                                 int *__restrict__ unsound) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes || where[i] < 0) return;
    const int kids[2] = {i + 1, nodes[i].offset};
    int slot[4];
    bool present[4];
    unsigned head_bits = 1u << nodes[i].axis;
    for (int k = 0; k < 2; ++k) {
        const int c = kids[k];
        if (nodes[c].n_prims != 0) {
            slot[2 * k] = c, present[2 * k] = true;
            slot[2 * k + 1] = 0, present[2 * k + 1] = false;
        } else {
            slot[2 * k] = c + 1, slot[2 * k + 1] = nodes[c].offset;
            present[2 * k] = present[2 * k + 1] = true;
            head_bits |= 1u << (3 + 3 * k + nodes[c].axis);
        }
    }
    unsigned ref[4];
    float lo_of[3][4], hi_of[3][4];
    for (int k = 0; k < 4; ++k) {
        if (!present[k]) {
            ref[k] = 0x40000000u;
            for (int a = 0; a < 3; ++a) lo_of[a][k] = hi_of[a][k] = 0.0f;
            continue;
        }
        const rt_bvh_node c = nodes[slot[k]];
        for (int a = 0; a < 3; ++a) lo_of[a][k] = c.bmin[a], hi_of[a][k] = c.bmax[a];
        if (!(c.bmin[0] <= c.bmax[0] && c.bmin[1] <= c.bmax[1] && c.bmin[2] <= c.bmax[2]) || isinf(c.bmin[0]) || isinf(c.bmin[1]) || isinf(c.bmin[2]) || isinf(c.bmax[0]) ||
            isinf(c.bmax[1]) || isinf(c.bmax[2])) {   // (a box that is not finite: the scene stays on the two-box records)
            *unsound = 1;
            return;
        }
        ref[k] = c.n_prims != 0 ? (0x80000000u | (unsigned)c.offset) : (unsigned)where[slot[k]];
    }
    if (!write_wide_record(lo_of, hi_of, present, ref, head_bits, wnodes + 4 * (size_t)where[i])) *unsound = 1;
}

}  // namespace

int build_bvh_on_device(const rt_bvh_prim *prims, const float *bounds, int n, const rt_triangle *d_triangles, int n_triangles, const rt_instance *d_instances,
                        float id_scale, int max_depth, int wide_stack_entries, DeviceBvh &out, std::string &err) {
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
    StageClock clock;
    clock.mark("boxes of the scene (host)");
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
    clock.mark("primitives uploaded");
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
    clock.mark("ordered along the curve");
    if (n == 1) {   // the root is the one leaf
        hipLaunchKernelGGL(entries_kernel, dim3(blocks_n), dim3(256), 0, 0, d_in, d_bounds, order, n, d_triangles, n_triangles, d_instances, id_scale, (const int *)nullptr,
                           (const int *)nullptr, (const int *)nullptr, (const int4 *)nullptr, (const float *)nullptr, d_prims, d_meta);
        BVH_HIP_OK(hipGetLastError());
        out.root_ref = (int)0x80000000u;
    } else {
        const int ni = n - 1;
        TreeArrays T;
        T.left = mem.get<int>((size_t)ni), T.right = mem.get<int>((size_t)ni), T.parent = mem.get<int>((size_t)ni), T.axis_bit = mem.get<int>((size_t)ni);
        T.info = mem.get<int4>((size_t)ni), T.box = mem.get<float>(6 * (size_t)ni), T.parent_of_leaf = mem.get<int>((size_t)n);
        T.live = mem.get<int>((size_t)ni);
        T.wide = mem.get<int4>((size_t)ni);
        int *record_of = mem.get<int>((size_t)ni), *wide_record_of = mem.get<int>((size_t)ni), *wide_top = mem.get<int>((size_t)ni);
        int *wide_queue = mem.get<int>(4 * (size_t)TOP_RECORDS + 4), *wide_words = mem.get<int>(2);   // {records of the top, 1 when a record could not be made}
        const size_t n_records = (size_t)TOP_RECORDS + (size_t)ni;   // (room for every node; the walk reaches the ones that are not inside a leaf)
        float4 *d_nodes = mem.get<float4>(4 * n_records);
        // clusters of the current and of the next round; a round's flags and their scan
        int *ref[2] = {mem.get<int>((size_t)n), mem.get<int>((size_t)n)};
        int4 *info[2] = {mem.get<int4>((size_t)n), mem.get<int4>((size_t)n)};
        float *cbox[2] = {mem.get<float>(6 * (size_t)n), mem.get<float>(6 * (size_t)n)};
        int *nearest = mem.get<int>((size_t)n), *place = mem.get<int>((size_t)n), *leaf_of = mem.get<int>((size_t)n), *first_entry = mem.get<int>((size_t)ni);
        unsigned long long *flags = mem.get<unsigned long long>((size_t)n + 1), *scanned = mem.get<unsigned long long>((size_t)n + 1);
        size_t scan_bytes = 0;
        BVH_HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flags, scanned, n + 1));
        void *scan_temp = mem.get<char>(scan_bytes);
        if (!T.left || !T.right || !T.parent || !T.axis_bit || !T.info || !T.box || !T.parent_of_leaf || !T.live || !T.wide || !record_of || !wide_record_of || !wide_top ||
            !wide_queue || !wide_words || !d_nodes || !ref[0] || !ref[1] || !info[0] || !info[1] ||
            !cbox[0] || !cbox[1] || !nearest || !place || !leaf_of || !first_entry || !flags || !scanned || !scan_temp) {
            err = "rt_scene_create: out of device memory for the tree build";
            return 10;
        }
        const unsigned blocks_i = (unsigned)((ni + 255) / 256);
        const int radius = getenv("RT_GPU_BVH_RADIUS") ? std::max(1, std::min(256, atoi(getenv("RT_GPU_BVH_RADIUS")))) : 8;   // (1M-triangle scene, node visits per ray against the host tree: radius 2 +9 %, 4 +6 %, 8 +3 %, 16 +6 %, 48 +8 %)
        clock.mark("buffers of the clustering");
        hipLaunchKernelGGL(ploc_leaves_kernel, dim3(blocks_n), dim3(256), 0, 0, d_bounds, order, n, ref[0], cbox[0], info[0]);
        int nc = n, cur = 0;
        int *state = mem.get<int>(2);
        if (!state) {
            err = "rt_scene_create: out of device memory for the tree build";
            return 10;
        }
        const int first_state[2] = {n, 0};
        BVH_HIP_OK(hipMemcpy(state, first_state, sizeof first_state, hipMemcpyHostToDevice));
        constexpr int ROUNDS_PER_LOOK = 4;   // (a round merges a third of the clusters or more: four rounds leave a fifth)
        while (nc > 1) {
            const int bound = nc;
            const unsigned blocks_c = (unsigned)((bound + 1 + 255) / 256);
            for (int round = 0; round < ROUNDS_PER_LOOK; ++round) {
                hipLaunchKernelGGL(ploc_nearest_kernel, dim3(blocks_c), dim3(256), 0, 0, cbox[cur], (const int *)state, radius, nearest);
                hipLaunchKernelGGL(ploc_flag_kernel, dim3(blocks_c), dim3(256), 0, 0, nearest, (const int *)state, bound, flags);
                BVH_HIP_OK(hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_bytes, flags, scanned, bound + 1));
                hipLaunchKernelGGL(ploc_merge_kernel, dim3(blocks_c), dim3(256), 0, 0, nearest, flags, scanned, (const int *)state, ref[cur], cbox[cur], info[cur], ref[cur ^ 1],
                                   cbox[cur ^ 1], info[cur ^ 1], T);
                hipLaunchKernelGGL(ploc_advance_kernel, dim3(1), dim3(64), 0, 0, scanned, bound, state);
                cur ^= 1;
            }
            BVH_HIP_OK(hipGetLastError());
            int now[2] = {0, 0};
            BVH_HIP_OK(hipMemcpy(now, state, sizeof now, hipMemcpyDeviceToHost));
            if (now[0] < 1 || now[0] >= nc || now[1] != n - now[0]) {   // (the best pair of a round always choose each other)
                err = "rt_scene_create: the clustering of the tree build made no progress";
                return 10;
            }
            nc = now[0];
        }
        clock.mark("clustered");
        int4 root_info;
        BVH_HIP_OK(hipMemcpy(&root_info, info[cur], sizeof root_info, hipMemcpyDeviceToHost));
        out.depth = root_info.x;
        if (out.depth > max_depth) {
            err = "rt_scene_create: the tree built on the device is " + std::to_string(out.depth) + " levels deep, more than the walk's stack of " +
                  std::to_string(max_depth) + " holds (many primitives with the same centre); build this scene's tree on the host";
            return 7;
        }
        hipLaunchKernelGGL(ploc_places_kernel, dim3(blocks_n), dim3(256), 0, 0, n, T, place, leaf_of, first_entry);
        hipLaunchKernelGGL(entries_kernel, dim3(blocks_n), dim3(256), 0, 0, d_in, d_bounds, order, n, d_triangles, n_triangles, d_instances, id_scale, place, leaf_of,
                           first_entry, T.info, T.box, d_prims, d_meta);
        BVH_HIP_OK(hipMemsetAsync(d_nodes, 0, 4 * n_records * sizeof(float4), 0));   // (places of the top that the tree does not fill are copied to LDS with the rest)
        // the four-box records as well (the walk of every frame that does not count its steps), unless a walk could need more
        // stack than the trace kernel has
        int4 root_wide = make_int4(0, 0, 0, 0);
        BVH_HIP_OK(hipMemcpy(&root_wide, T.wide + (ni - 1), sizeof root_wide, hipMemcpyDeviceToHost));   // (the root is the node made last)
        const bool wide = wide_stack_entries > 0 && !root_info.w && root_wide.z <= wide_stack_entries;
        float4 *d_wnodes = nullptr;
        if (wide) {
            d_wnodes = mem.get<float4>(4 * ((size_t)TOP_RECORDS + (size_t)root_wide.x));
            if (!d_wnodes) {
                err = "rt_scene_create: out of device memory for the tree build";
                return 10;
            }
            BVH_HIP_OK(hipMemsetAsync(d_wnodes, 0, 4 * ((size_t)TOP_RECORDS + (size_t)root_wide.x) * sizeof(float4), 0));
            BVH_HIP_OK(hipMemsetAsync(wide_top, 0xff, (size_t)ni * sizeof(int), 0));
            BVH_HIP_OK(hipMemsetAsync(wide_words, 0, 2 * sizeof(int), 0));
            hipLaunchKernelGGL(wide_top_kernel, dim3(1), dim3(64), 0, 0, ni - 1, T, wide_top, wide_queue, wide_words);
        }
        hipLaunchKernelGGL(ploc_record_places_kernel, dim3(blocks_i), dim3(256), 0, 0, ni, T, (const int *)wide_top, record_of, wide ? wide_record_of : (int *)nullptr);
        hipLaunchKernelGGL(ploc_records_kernel, dim3(blocks_i), dim3(256), 0, 0, d_bounds, order, ni, T, place, first_entry, record_of, d_nodes);
        if (wide)
            hipLaunchKernelGGL(ploc_wide_records_kernel, dim3(blocks_i), dim3(256), 0, 0, d_bounds, order, ni, T, place, first_entry, (const int *)wide_record_of, d_wnodes,
                               wide_words + 1);
        BVH_HIP_OK(hipGetLastError());
        if (root_info.w) {   // at most four primitives in all, cheaper as one leaf: the root is that leaf
            out.root_ref = (int)0x80000000u;
        } else {
            out.nodes = d_nodes;
            out.n_interior = ni;
            out.n_top = TOP_RECORDS;
            out.root_ref = 0;
            mem.release(d_nodes);
            if (wide) {
                int words[2] = {0, 1};
                BVH_HIP_OK(hipMemcpy(words, wide_words, sizeof words, hipMemcpyDeviceToHost));
                if (!words[1]) {   // (a record that could not be made leaves the scene on the two-box records)
                    out.wide_nodes = d_wnodes;
                    out.n_wide = TOP_RECORDS + root_wide.x;
                    out.n_wide_heads = root_wide.x;
                    out.n_wide_top = words[0];
                    mem.release(d_wnodes);
                }
            }
        }
    }
    BVH_HIP_OK(hipDeviceSynchronize());
    clock.mark("entries and records written");
    out.prims = d_prims;
    out.prim_meta = d_meta;
    mem.release(d_prims);
    mem.release(d_meta);
    return 0;
}

int records_of_host_tree(const rt_bvh_node *nodes, int n_nodes, const int *top, int n_top, int n_interior, const rt_bvh_prim *prims, int n, const rt_triangle *d_triangles,
                         int n_triangles, const rt_instance *d_instances, float id_scale, bool keep_nodes, DeviceBvh &out, std::string &err) {
    out = DeviceBvh();
    if (n_top < 0 || n_top > TOP_RECORDS || n_interior < n_top) {
        err = "rt_scene_create: bad list of the tree's top nodes";
        return 3;
    }
    Scratch mem;
    // the top nodes' records take the first n_top places; every interior node's number in array order, behind them, is its
    // place otherwise (the places of the top nodes in that numbering stay empty)
    const int n_records = n_interior > 0 ? n_top + n_interior : 0;
    rt_bvh_node *d_nodes_in = mem.get<rt_bvh_node>((size_t)n_nodes);
    int *d_record_of = mem.get<int>((size_t)n_nodes), *d_interior = mem.get<int>((size_t)n_nodes), *d_rank = mem.get<int>((size_t)n_nodes), *d_top = mem.get<int>((size_t)TOP_RECORDS);
    rt_bvh_prim *d_in = mem.get<rt_bvh_prim>((size_t)n);
    float4 *d_records = mem.get<float4>(4 * (size_t)n_records), *d_prims = mem.get<float4>(PRIM_STRIDE * (size_t)n);
    int4 *d_meta = mem.get<int4>(2 * (size_t)n);
    size_t scan_bytes = 0;
    BVH_HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_interior, d_rank, std::max(n_nodes, 1)));
    void *scan_temp = mem.get<char>(scan_bytes);
    if (!d_nodes_in || !d_record_of || !d_interior || !d_rank || !d_top || !d_in || !d_records || !d_prims || !d_meta || !scan_temp) {
        err = "rt_scene_create: out of device memory for the tree's records";
        return 10;
    }
    if (n_nodes > 0) BVH_HIP_OK(hipMemcpy(d_nodes_in, nodes, (size_t)n_nodes * sizeof(rt_bvh_node), hipMemcpyHostToDevice));
    if (n_top > 0) BVH_HIP_OK(hipMemcpy(d_top, top, (size_t)n_top * sizeof(int), hipMemcpyHostToDevice));
    if (n > 0) BVH_HIP_OK(hipMemcpy(d_in, prims, (size_t)n * sizeof(rt_bvh_prim), hipMemcpyHostToDevice));
    if (n > 0)
        hipLaunchKernelGGL(entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_in, (const float *)nullptr, (const unsigned *)nullptr, n, d_triangles, n_triangles,
                           d_instances, id_scale, (const int *)nullptr, (const int *)nullptr, (const int *)nullptr, (const int4 *)nullptr, (const float *)nullptr, d_prims, d_meta);
    if (n_nodes > 0) {
        const unsigned blocks = (unsigned)((n_nodes + 255) / 256);
        hipLaunchKernelGGL(leaf_boxes_kernel, dim3(blocks), dim3(256), 0, 0, d_nodes_in, n_nodes, d_prims);
        if (n_records > 0) {
            BVH_HIP_OK(hipMemsetAsync(d_records, 0, 4 * (size_t)n_records * sizeof(float4), 0));
            hipLaunchKernelGGL(tree_interior_kernel, dim3(blocks), dim3(256), 0, 0, (const rt_bvh_node *)d_nodes_in, n_nodes, d_interior);
            BVH_HIP_OK(hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_bytes, d_interior, d_rank, n_nodes));
            hipLaunchKernelGGL(tree_places_kernel, dim3(blocks), dim3(256), 0, 0, n_nodes, (const int *)d_interior, (const int *)d_rank, n_top, d_record_of);
            hipLaunchKernelGGL(tree_top_places_kernel, dim3(1), dim3(128), 0, 0, (const int *)d_top, n_top, d_record_of);
            hipLaunchKernelGGL(node_records_kernel, dim3(blocks), dim3(256), 0, 0, d_nodes_in, n_nodes, d_record_of, d_records);
        }
    }
    BVH_HIP_OK(hipGetLastError());
    BVH_HIP_OK(hipDeviceSynchronize());
    out.nodes = d_records, out.prims = d_prims, out.prim_meta = d_meta;
    out.n_interior = n_records;
    if (keep_nodes) {
        out.tree_nodes = d_nodes_in;
        mem.release(d_nodes_in);
    }
    mem.release(d_records);
    mem.release(d_prims);
    mem.release(d_meta);
    return 0;
}

int wide_records_of_host_tree(const rt_bvh_node *d_nodes, int n_nodes, const int *top, int n_top, int stack_entries, DeviceBvh &out, std::string &err) {
    out.wide_nodes = nullptr, out.n_wide = 0, out.n_wide_top = 0;
    if (n_nodes < 3 || n_top < 1 || n_top > TOP_RECORDS) return 0;
    Scratch mem;
    int *parent = mem.get<int>((size_t)n_nodes), *head = mem.get<int>((size_t)n_nodes + 1), *rank = mem.get<int>((size_t)n_nodes + 1), *where = mem.get<int>((size_t)n_nodes);
    int *d_top = mem.get<int>((size_t)TOP_RECORDS), *words = mem.get<int>(2);   // {stack a walk needs, 1 when a record could not be made}
    size_t scan_bytes = 0;
    BVH_HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, head, rank, n_nodes + 1));
    void *scan_temp = mem.get<char>(scan_bytes);
    if (!parent || !head || !rank || !where || !d_top || !words || !scan_temp) {
        err = "out of device memory for the wide tree";
        return 10;
    }
    const unsigned blocks = (unsigned)((n_nodes + 255) / 256);
    BVH_HIP_OK(hipMemcpy(d_top, top, (size_t)n_top * sizeof(int), hipMemcpyHostToDevice));
    BVH_HIP_OK(hipMemsetAsync(words, 0, 2 * sizeof(int), 0));
    BVH_HIP_OK(hipMemsetAsync(head + n_nodes, 0, sizeof(int), 0));   // the scan's last element: the number of heads
    hipLaunchKernelGGL(tree_parents_kernel, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, parent);
    hipLaunchKernelGGL(tree_heads_kernel, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, (const int *)parent, head, words);
    BVH_HIP_OK(hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_bytes, head, rank, n_nodes + 1));
    BVH_HIP_OK(hipGetLastError());
    int n_heads = 0, need = 0;
    BVH_HIP_OK(hipMemcpy(&n_heads, rank + n_nodes, sizeof n_heads, hipMemcpyDeviceToHost));
    BVH_HIP_OK(hipMemcpy(&need, words, sizeof need, hipMemcpyDeviceToHost));
    if (n_heads < 1 || need > stack_entries) return 0;   // (a walk could need more stack than the trace kernel has: two-box records)
    const size_t n_records = (size_t)TOP_RECORDS + (size_t)n_heads;
    float4 *d_wnodes = mem.get<float4>(4 * n_records);
    if (!d_wnodes) {
        err = "out of device memory for the wide tree";
        return 10;
    }
    BVH_HIP_OK(hipMemsetAsync(d_wnodes, 0, 4 * n_records * sizeof(float4), 0));
    hipLaunchKernelGGL(tree_places_kernel, dim3(blocks), dim3(256), 0, 0, n_nodes, (const int *)head, (const int *)rank, TOP_RECORDS, where);
    hipLaunchKernelGGL(tree_top_places_kernel, dim3(1), dim3(128), 0, 0, (const int *)d_top, n_top, where);
    hipLaunchKernelGGL(tree_wide_records_kernel, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, (const int *)where, d_wnodes, words + 1);
    BVH_HIP_OK(hipGetLastError());
    int unsound = 1;
    BVH_HIP_OK(hipMemcpy(&unsound, words + 1, sizeof unsound, hipMemcpyDeviceToHost));
    if (unsound) return 0;
    out.wide_nodes = d_wnodes;
    out.n_wide = (int)n_records;
    out.n_wide_heads = n_heads;
    out.n_wide_top = n_top;
    mem.release(d_wnodes);
    return 0;
}

}  // namespace rt
