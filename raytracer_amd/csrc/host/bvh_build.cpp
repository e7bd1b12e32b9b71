// SAH BVH construction on the host (reference src/bvh.cpp:41-344).
//
// 12-bucket surface-area heuristic on the axis of largest centroid extent, traversal cost 0.125,
// a leaf when n <= 4 and no split is cheaper than n, an equal-count split when the SAH partition
// degenerates.  Nodes are emitted directly in depth-first pre-order, which is the layout the
// reference produces with its separate flatten pass (left child = i+1, right child = offset).
// std::partition / std::nth_element are used exactly where the reference uses them so that the
// primitive order inside every subtree — and with it the order in which equal-distance hits are
// found — is the same.
//
// The reference builds on one thread.  Here the top of the tree is built on the calling thread and every
// subtree of at most SUBTREE_ITEMS primitives becomes a task for a pool of threads: a subtree only ever
// touches its own range of the item array, its leaf offsets are the range's own positions (leaves are
// emitted in array order), and its node indices are relative until the subtrees are laid out one after
// the other in pre-order.  The result is the same array, byte for byte, for any number of threads.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "scene.h"

namespace rth {

namespace {

struct BuildItem {
    int prim;
    float bmin[3], bmax[3], c[3];
};

constexpr int SUBTREE_ITEMS = 32768;

struct Subtree {
    int start, end;
    std::vector<rt_bvh_node> nodes;   // pre-order, child offsets relative to this vector
};

struct Builder {
    BuildItem *items = nullptr;           // shared array; a builder only touches its own range
    std::vector<rt_bvh_node> nodes;       // pre-order; offset of a deferred subtree's placeholder = -(1 + subtree number)
    std::vector<Subtree> *deferred = nullptr;   // set on the top-level builder only

    // Half the surface area of a box, in the arithmetic of the reference's build: its Makefile's -ffast-math lets the
    // compiler turn x y + x z + y z (src/bvh.cpp:249-257) into one fused x (y + z) + y z, and a split's cost (:259) into
    // fma(fma(S1, 2 n1, (2 n0) S0), 1 / (2 S), 0.125).  The last bit matters wherever two splits cost the same on paper --
    // the two opposite faces of any axis-aligned box -- because the first strictly cheaper split wins (:262-271).
    static float half_area(const float *mn, const float *mx) {
        float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
        if (x < 0 || y < 0 || z < 0) return 0.0f;
        return fmaf(x, y + z, y * z);
    }

    void make_leaf(int node, int start, int end) {
        nodes[node].offset = start;   // leaves are emitted in array order, so the running count equals `start`
        nodes[node].n_prims = end - start;
        nodes[node].axis = 0;
    }

    int build(int start, int end) {
        const int node = (int)nodes.size();
        nodes.emplace_back();
        if (deferred && end - start <= SUBTREE_ITEMS && end - start > 1) {   // a task for the pool
            nodes[node].n_prims = -1;
            nodes[node].offset = -(1 + (int)deferred->size());
            deferred->push_back(Subtree{start, end, {}});
            return node;
        }
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = start; i < end; ++i)
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], items[i].bmin[a]);
                mx[a] = std::max(mx[a], items[i].bmax[a]);
            }
        memcpy(nodes[node].bmin, mn, sizeof mn);
        memcpy(nodes[node].bmax, mx, sizeof mx);

        const int n = end - start;
        if (n == 1) {
            make_leaf(node, start, end);
            return node;
        }
        float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = start; i < end; ++i)
            for (int a = 0; a < 3; ++a) {
                cmn[a] = std::min(cmn[a], items[i].c[a]);
                cmx[a] = std::max(cmx[a], items[i].c[a]);
            }
        int dim = 0;
        float extent = cmx[0] - cmn[0];
        if ((cmx[1] - cmn[1]) > extent) {
            dim = 1;
            extent = cmx[1] - cmn[1];
        }
        if ((cmx[2] - cmn[2]) > extent) dim = 2;   // the reference leaves `extent` at the x/y value here
        if (extent < 1e-5f) {
            make_leaf(node, start, end);
            return node;
        }

        constexpr int NB = 12;
        struct Bucket {
            int count = 0;
            float mn[3], mx[3];
        } buckets[NB];
        const float k1 = NB * (1.0f - 1e-4f) / extent;
        const float cmin = cmn[dim];
        auto bucket_of = [=](const BuildItem &it) {
            int b = (int)((it.c[dim] - cmin) * k1);
            if (b < 0) b = 0;
            if (b >= NB) b = NB - 1;
            return b;
        };
        for (int i = start; i < end; ++i) {
            Bucket &b = buckets[bucket_of(items[i])];
            if (b.count++ == 0) {
                memcpy(b.mn, items[i].bmin, sizeof b.mn);
                memcpy(b.mx, items[i].bmax, sizeof b.mx);
            } else {
                for (int a = 0; a < 3; ++a) {
                    b.mn[a] = std::min(b.mn[a], items[i].bmin[a]);
                    b.mx[a] = std::max(b.mx[a], items[i].bmax[a]);
                }
            }
        }
        float cost[NB - 1];
        const float parent_half = half_area(mn, mx);
        const float inv_parent_sa = 1.0f / (parent_half + parent_half);
        for (int i = 0; i < NB - 1; ++i) {
            float mn0[3] = {0, 0, 0}, mx0[3] = {0, 0, 0}, mn1[3] = {0, 0, 0}, mx1[3] = {0, 0, 0};
            int c0 = 0, c1 = 0;
            bool init0 = false, init1 = false;
            auto grow = [](float *dmn, float *dmx, bool &init, const Bucket &b) {
                if (!init) {
                    memcpy(dmn, b.mn, 12);
                    memcpy(dmx, b.mx, 12);
                    init = true;
                } else {
                    for (int a = 0; a < 3; ++a) {
                        dmn[a] = std::min(dmn[a], b.mn[a]);
                        dmx[a] = std::max(dmx[a], b.mx[a]);
                    }
                }
            };
            for (int j = 0; j <= i; ++j)
                if (buckets[j].count) {
                    c0 += buckets[j].count;
                    grow(mn0, mx0, init0, buckets[j]);
                }
            for (int j = i + 1; j < NB; ++j)
                if (buckets[j].count) {
                    c1 += buckets[j].count;
                    grow(mn1, mx1, init1, buckets[j]);
                }
            const float n0 = (float)c0, n1 = (float)c1;
            cost[i] = fmaf(fmaf(half_area(mn1, mx1), n1 + n1, (n0 + n0) * half_area(mn0, mx0)), inv_parent_sa, 0.125f);
        }
        float min_cost = cost[0];
        int split_bucket = 0;
        for (int i = 1; i < NB - 1; ++i)
            if (cost[i] < min_cost) {
                min_cost = cost[i];
                split_bucket = i;
            }
        if (n > 4 || min_cost < (float)n) {
            auto pmid = std::partition(items + start, items + end, [=](const BuildItem &it) { return bucket_of(it) <= split_bucket; });
            int mid = (int)(pmid - items);
            if (mid == start || mid == end) {
                mid = start + (end - start) / 2;
                std::nth_element(items + start, items + mid, items + end,
                                 [dim](const BuildItem &a, const BuildItem &b) { return a.c[dim] < b.c[dim]; });
            }
            nodes[node].axis = dim;
            nodes[node].n_prims = 0;
            build(start, mid);
            int second = build(mid, end);
            nodes[node].offset = second;
        } else {
            make_leaf(node, start, end);
        }
        return node;
    }
};

}  // namespace

void build_bvh(HostScene &scene) {
    scene.bvh_nodes.clear();
    scene.bvh_prims.clear();
    const size_t n = scene.build_prims.size();
    if (n == 0) return;
    std::vector<BuildItem> items(n);
    for (size_t i = 0; i < n; ++i) {
        BuildItem &it = items[i];
        it.prim = (int)i;
        memcpy(it.bmin, scene.build_bounds[i].bmin, 12);
        memcpy(it.bmax, scene.build_bounds[i].bmax, 12);
        for (int a = 0; a < 3; ++a) it.c[a] = (it.bmin[a] + it.bmax[a]) * 0.5f;
    }
    // top of the tree on this thread; subtrees of <= SUBTREE_ITEMS primitives are deferred
    std::vector<Subtree> subtrees;
    Builder top;
    top.items = items.data();
    top.deferred = &subtrees;
    top.build(0, (int)n);
    // the deferred subtrees on a pool of threads, largest first
    std::vector<int> by_size(subtrees.size());
    for (size_t i = 0; i < by_size.size(); ++i) by_size[i] = (int)i;
    std::sort(by_size.begin(), by_size.end(), [&](int a, int b) {
        const int sa = subtrees[a].end - subtrees[a].start, sb = subtrees[b].end - subtrees[b].start;
        return sa != sb ? sa > sb : a < b;
    });
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= by_size.size()) return;
            Subtree &st = subtrees[by_size[k]];
            Builder b;
            b.items = items.data();
            b.nodes.reserve(2 * (size_t)(st.end - st.start));
            b.build(st.start, st.end);
            st.nodes = std::move(b.nodes);
        }
    };
    unsigned n_threads = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    n_threads = (unsigned)std::min<size_t>(n_threads, std::max<size_t>(1, subtrees.size()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    // lay the pieces out in pre-order: final index of every top node, then copy with the offsets shifted
    std::vector<int> final_index(top.nodes.size());
    size_t total = 0;
    for (size_t i = 0; i < top.nodes.size(); ++i) {
        final_index[i] = (int)total;
        total += top.nodes[i].n_prims == -1 ? subtrees[-top.nodes[i].offset - 1].nodes.size() : 1;
    }
    scene.bvh_nodes.resize(total);
    for (size_t i = 0; i < top.nodes.size(); ++i) {
        const rt_bvh_node &t = top.nodes[i];
        rt_bvh_node *dst = &scene.bvh_nodes[final_index[i]];
        if (t.n_prims == -1) {
            const Subtree &st = subtrees[-t.offset - 1];
            const int base = final_index[i];
            for (size_t k = 0; k < st.nodes.size(); ++k) {
                dst[k] = st.nodes[k];
                if (dst[k].n_prims == 0) dst[k].offset += base;   // interior: index of the second child
            }
        } else {
            *dst = t;
            if (t.n_prims == 0) dst->offset = final_index[t.offset];
        }
    }
    scene.bvh_prims.resize(n);
    for (size_t i = 0; i < n; ++i) scene.bvh_prims[i] = scene.build_prims[items[i].prim];
}

}  // namespace rth
