// Host-side construction of the wide tree the trace kernel walks (DevScene::wnodes in rt_device.h describes the
// record).  Plain C++: rt_scene_create uses it, and the CPU checker builds the same records to walk them on the host.
#pragma once
#include <cmath>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "gpu_render.h"

namespace rt {

struct WideRecord {
    float f[16];
};
constexpr int WIDE_REF_NONE = 0x40000000;   // slot without a box: the walk's "pop" reference, harmless if ever followed

// The four slots of the record that node i heads (-1: empty) and the split axes of i and of its interior children.
inline void wide_slots(const rt_bvh_node *N, int i, int slot[4], int axis_of[3]) {
    const int L = i + 1, R = N[i].offset;
    axis_of[0] = N[i].axis, axis_of[1] = axis_of[2] = -1;
    if (N[L].n_prims == 0) slot[0] = L + 1, slot[1] = N[L].offset, axis_of[1] = N[L].axis; else slot[0] = L, slot[1] = -1;
    if (N[R].n_prims == 0) slot[2] = R + 1, slot[3] = N[R].offset, axis_of[2] = N[R].axis; else slot[2] = R, slot[3] = -1;
}

// Fills the record headed by node i (`index`: node -> record); false when a box is not finite or the grid cannot hold
// the boxes.
inline bool fill_wide_record(const rt_bvh_node *N, int i, const std::vector<int> &index, WideRecord &rec_out) {
    auto as_float = [](unsigned v) {
        float f;
        memcpy(&f, &v, 4);
        return f;
    };
    int slot[4], ax[3];
    wide_slots(N, i, slot, ax);
    unsigned ref[4], qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    float origin[3], scale[3];
    for (int k = 0; k < 4; ++k) {
        if (slot[k] < 0) {
            ref[k] = (unsigned)WIDE_REF_NONE;
            continue;
        }
        const rt_bvh_node &c = N[slot[k]];
        if (c.n_prims == 0) {
            ref[k] = (unsigned)index[slot[k]];
        } else {
            ref[k] = 0x80000000u | (unsigned)c.offset;   // the leaf's first entry; its last entry carries the end mark
        }
    }
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < 4; ++k) {
            if (slot[k] < 0) continue;
            const rt_bvh_node &c = N[slot[k]];
            if (!std::isfinite(c.bmin[a]) || !std::isfinite(c.bmax[a]) || c.bmin[a] > c.bmax[a]) return false;
            lo = std::min(lo, c.bmin[a]);
            hi = std::max(hi, c.bmax[a]);
        }
        origin[a] = lo;
        // the grid step is never so small that origin + 255 steps rounds back to the origin: an empty slot's
        // inverted box (255..0) then misses on every axis the ray moves along
        int e = std::max(-100, std::ilogb(std::max(std::max(std::fabs(lo), std::fabs(hi)), 1e-30f)) - 22);
        const double extent = (double)hi - (double)lo;
        if (extent > 0) e = std::max(std::ilogb(extent / 255.0) + 1, e);   // 2^e > extent / 255
        for (;; ++e) {   // grid step 2^e: the smallest with which every box fits 0..255
            if (e > 126) return false;
            const float step = std::ldexp(1.0f, e);
            const double per_step = std::ldexp(1.0, -e);   // a power of two: the products below are exact
            unsigned lo_bytes = 0, hi_bytes = 0;
            bool fits = true;
            for (int k = 0; k < 4 && fits; ++k) {
                if (slot[k] < 0) {   // an empty slot keeps the inverted box 255..0 (and its ref says so)
                    lo_bytes |= 255u << (8 * k);
                    continue;
                }
                const rt_bvh_node &c = N[slot[k]];
                // the decoded coordinate is fmaf(q, step, origin), on the device too: never inside the exact box
                int ql = (int)std::min(255.0, std::floor(((double)c.bmin[a] - (double)lo) * per_step));
                while (ql > 0 && fmaf((float)ql, step, lo) > c.bmin[a]) --ql;
                int qh = (int)std::min(256.0, std::ceil(((double)c.bmax[a] - (double)lo) * per_step));
                while (qh <= 255 && fmaf((float)qh, step, lo) < c.bmax[a]) ++qh;
                if (qh > 255) fits = false;
                lo_bytes |= (unsigned)ql << (8 * k);
                hi_bytes |= (unsigned)(qh & 255) << (8 * k);
            }
            if (fits) {
                qlo[a] = lo_bytes, qhi[a] = hi_bytes;
                scale[a] = step;
                break;
            }
        }
    }
    unsigned head;
    memcpy(&head, &scale[0], 4);
    head |= (1u << ax[0]) | (ax[1] >= 0 ? 1u << (3 + ax[1]) : 0u) | (ax[2] >= 0 ? 1u << (6 + ax[2]) : 0u);
    const float rec[16] = {origin[0], origin[1], origin[2], as_float(head), as_float(ref[0]), as_float(ref[1]), as_float(ref[2]), as_float(ref[3]),
                           as_float(qlo[0]), as_float(qhi[0]), as_float(qlo[1]), as_float(qhi[1]), as_float(qlo[2]), as_float(qhi[2]), scale[1], scale[2]};
    memcpy(rec_out.f, rec, sizeof rec);
    return true;
}

// Builds the wide tree (DevScene::wnodes) from the two-child tree: node i, its interior children and their children
// become one record.  Returns false (and the scene renders from the two-box records) when the tree does not qualify:
// a single leaf, a box that is not finite, or a walk that could need more than the 64 stack entries.
inline bool build_wide_nodes(const rt_bvh_node *N, int n_nodes, int top_nodes, int stack_entries, std::vector<WideRecord> &out, int &root_ref, int &n_top) {
    if (n_nodes < 3 || N[0].n_prims != 0) return false;
    // record order: the first TOP_NODES breadth-first, every subtree below depth-first
    std::vector<int> index(n_nodes, -1), heads;
    {
        std::vector<int> walk{0}, below, stack;
        for (size_t at = 0; at < walk.size(); ++at) {
            const int i = walk[at];
            if ((int)heads.size() >= top_nodes) {
                below.push_back(i);
                continue;
            }
            index[i] = (int)heads.size();
            heads.push_back(i);
            int slot[4], ax[3];
            wide_slots(N, i, slot, ax);
            for (int k = 0; k < 4; ++k)
                if (slot[k] >= 0 && N[slot[k]].n_prims == 0) walk.push_back(slot[k]);
        }
        n_top = (int)heads.size();
        for (int root : below) {
            stack.push_back(root);
            while (!stack.empty()) {
                const int i = stack.back();
                stack.pop_back();
                index[i] = (int)heads.size();
                heads.push_back(i);
                int slot[4], ax[3];
                wide_slots(N, i, slot, ax);
                for (int k = 3; k >= 0; --k)
                    if (slot[k] >= 0 && N[slot[k]].n_prims == 0) stack.push_back(slot[k]);
            }
        }
    }
    // stack entries a walk can hold below each head: its waiting slots plus the deepest child's need (children first:
    // `heads` lists every node after its parent)
    {
        std::vector<int> need(heads.size(), 0);
        for (size_t h = heads.size(); h-- > 0;) {
            int slot[4], ax[3], present = 0, deepest = 0;
            wide_slots(N, heads[h], slot, ax);
            for (int k = 0; k < 4; ++k) {
                if (slot[k] < 0) continue;
                ++present;
                if (N[slot[k]].n_prims == 0) deepest = std::max(deepest, need[(size_t)index[slot[k]]]);
            }
            need[h] = present - 1 + deepest;
        }
        if (need[0] > stack_entries) return false;
    }
    // the records are independent of each other: filled by a few threads (a scene of 1M triangles has 430 K of them)
    out.assign(heads.size(), WideRecord{});
    std::atomic<bool> sound(true);
    auto fill = [&](size_t h_begin, size_t h_end) {
        for (size_t h = h_begin; h < h_end; ++h)
            if (!fill_wide_record(N, heads[h], index, out[h])) sound = false;
    };
    const size_t n_threads = std::min<size_t>(std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16), heads.size() / 8192 + 1);
    if (n_threads <= 1) {
        fill(0, heads.size());
    } else {
        std::vector<std::thread> pool;
        const size_t share = (heads.size() + n_threads - 1) / n_threads;
        for (size_t t = 0; t < n_threads; ++t) pool.emplace_back(fill, std::min(heads.size(), t * share), std::min(heads.size(), (t + 1) * share));
        for (auto &th : pool) th.join();
    }
    if (!sound) return false;
    root_ref = 0;
    return true;
}


}  // namespace rt
