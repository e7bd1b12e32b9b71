// MI355X (gfx950, wave64) rendering core: C ABI of include/gpu_render.h and the HIP kernels behind it.
//
// What is replaced: the reference's per-camera std::thread scheduler and everything it calls at
// render time (reference src/main.cpp:437-555 -> camera::get_samples src/camera.cpp:117-224,
// ray_tracer::trace src/ray_trace.cpp:1141-1172, bvh::intersect src/bvh.cpp:383-480, the shape
// tests src/shapes.cpp:212,407,506, calculate_color src/ray_trace.cpp:154-406, Light::get_sample
// src/lights.cpp, texture::sample src/texture.cpp:168-197).
//
// Execution shape: a wavefront pipeline.  A frame is rendered in batches of camera samples; per batch the paths
// advance one generation at a time through generate -> trace (closest hit) -> shade, and the vertices of a
// generation are lit by light_setup -> trace (any hit) -> light_gather.  Queues are compacted
// with wave ballots and split into 64 shards; frames shard over ranks by 32x32 image tiles (the reference's
// square_dynamic tile, src/main.cpp:441).  A slot's random numbers depend only on (seed, pixel, sample, path node),
// so results do not depend on launch geometry, batching, sharding or scheduling.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "gpu_render.h"
#include "rt_math.h"
#include "rt_device.h"
#include "rt_wide.h"
#include "rt_bvh_gpu.h"
#include "rt_devmem.h"

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

// =====================================================================================================
// Wavefront pipeline.
//
// A frame is rendered in batches of camera samples ("slots").  Per batch:
//   generate  camera rays of the batch's (pixel, sample) pairs              -> path queue 0
//   repeat    closest   closest-hit BVH traversal of every queued path       -> hit records
//             shade     material / texture / emission / roulette; appends direct-lighting work
//                       (vertex records) and child paths, compacted with wave ballots -> path queue 1
//             direct    per vertex: shadow rays and light-sample records (light_setup), any-hit shadow
//                       traversal, then the sum over the light samples in sample order (light_gather)
//   resolve   per pixel: clamp each sample, sum the samples in index order, divide
// The recursion of the reference (trace_rec / path_trace) becomes the loop over queue generations; a
// path carries its throughput, so the order in which paths are processed does not change what is
// added to a slot, and a slot's random numbers depend only on (seed, pixel, sample, path node).
// =====================================================================================================

// waves per SIMD of the trace kernel, measured (closest-hit time per 4K / 64 spp frame of the 1M-triangle scene).  With the
// two-box records 5 was fastest (4: 475 ms, 5: 426, 6: 438, 7: 459, 8: 467: more waves evict each other's part of the tree
// from the caches); the wide tree moves less data per ray and gains from a sixth wave (4: 307 ms, 5: 273, 6: 255), which
// is what 82 registers and 26 KB of LDS per workgroup (stack 22 KB + top of the tree 4 KB) allow.  A seventh (72 registers, no
// spills; ten stack levels and 31 top records in LDS: 22 KB per workgroup) pays a little once the steps of an iteration share
// it: four alternating runs 228.9 - 230.2 ms against 225.4 - 226.3 (with 63 top records and nine levels 228.1, no gain).
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 7
#endif
// Queue generations a batch can have: path depths 0 .. ceil(MaxRecursionDepth), plus the children the last one appends.
// The depth travels in 16 bits of a path record; the append counters are allocated for what the scene asks for.
constexpr int MAX_DEPTH = 65000;

struct PathQueue {
    float4 *o_time;     // origin.xyz, ray time
    float4 *d_key;      // direction.xyz, bits(path-node key)
    float4 *thr_slot;   // throughput.xyz, bits(slot)
    int *depth_flags;   // depth | culling << 16 | diffuse bounce << 17
};

struct VertexBuf {      // one record per shading point that needs direct lighting
    float4 *p_time;     // position.xyz, ray time
    float4 *n_key;      // shading normal.xyz, bits(key)
    float4 *view_mat;   // unit vector towards the ray origin .xyz, bits(BRDF selector of the material, see brdf_code)
    float4 *kd_slot;    // diffuse reflectance after textures .xyz, bits(slot | 1 << 31 when the 0.004 LDR scale applies, src/ray_trace.cpp:402-405)
    float4 *ks_n;       // specular reflectance after textures .xyz, Phong exponent of the material
    float4 *weight;     // path throughput .xyz, bits(samples 32..63 of this pass that have a record)
    float4 *amb_first;  // ambient term .xyz (zero when calculate_color excludes it), bits(samples 0..31 that have a record)
};

struct ShadowQueue {
    // one ray per DISTINCT shadow ray of a vertex (light samples whose origin, direction and length agree bit for
    // bit share one: e.g. the 16 samples of a sphere light, whose sampled position is always the sphere's centre)
    float4 *o_dist;     // shadow-ray origin.xyz, distance to the light sample
    float4 *d_time;     // direction.xyz, ray time
    float *vis;         // visibility, written by the any-hit trace
    // one record per recorded light sample with non-zero incident radiance; for a deferred emitter, .w of the record of
    // a block's lane 0 holds the index of the block's first ray to that emitter (see light_setup_kernel)
    float4 *fd_ray;     // diffuse term x radiance x 1/samples .xyz, bits(index of the sample's ray)
    float4 *fs_;        // specular term x radiance x 1/samples .xyz
};

struct Wavefront {
    PathQueue q[2];
    float2 *hit;              // per path entry: closest distance, bits(leaf entry of the hit; <= -2 plane, -1 none, -3 depth limit): one 8-byte store per finished ray
    VertexBuf vtx[2];         // by generation parity: the direct lighting of generation g runs beside the
    ShadowQueue shadow[2];    // closest-hit trace and shading of generation g + 1
    unsigned shadow_shard_entries;   // shadow rays, and light-sample records, per shard
    int has_records;                 // the light-sample records (fd_ray, fs_) are allocated: a frame whose only light is one
                                     // sphere emitter evaluated after its ray needs none (64 of an entry's 484 bytes)
    float *accum;             // radiance sums per slot: three planes (r, g, b) of `capacity` floats
    unsigned *counters;       // [3 kinds][generations][N_SHARDS] entry counts, one 128-byte line each, then the overflow flag
    int generations;          // generations the counters have room for
    unsigned *overflow;       // set when a queue ran out of room
    unsigned shard_entries;   // path / vertex entries per shard
    unsigned capacity;        // N_SHARDS * shard_entries = slots of the accumulator
};

// Every queue is split into N_SHARDS regions with their own append counters: a producer workgroup appends to
// shard (blockIdx.x % N_SHARDS), a consumer workgroup walks the entries of shard (blockIdx.x % N_SHARDS) together
// with the other gridDim.x / N_SHARDS workgroups of that shard.  One counter for a whole queue would serialise
// millions of wave-level atomics on a single address.
constexpr int N_SHARDS = 64;
constexpr int COUNTER_STRIDE = 32;   // in 4-byte words
enum { Q_PATH = 0, Q_VERTEX = 1, Q_SHADOW = 2 };
__host__ __device__ __forceinline__ unsigned *counter_of(const Wavefront &W, int kind, int gen, int shard) {
    return W.counters + ((size_t)(kind * W.generations + gen) * N_SHARDS + shard) * COUNTER_STRIDE;
}

struct Batch {
    int tile_first, tile_count;   // range in this rank's list of owned tiles
    int n_pixels;                 // tile_count * 1024 (pixel-local indices, including out-of-image ones)
    int n_samples, grid_n;
    int tiles_x;
    const int *owned;             // this rank's tiles (row-major tile numbers, ascending); null: every tile, in order
    int still;                    // 1: rays carry time 0 whatever the camera sampled (primary-visibility queries)
};

__device__ __forceinline__ void batch_pixel(const Batch &b, int pl, int &i, int &j) {
    const int at = b.tile_first + (pl >> 10);
    const int tile = b.owned ? b.owned[at] : at;
    const int tx = tile % b.tiles_x, ty = tile / b.tiles_x;
    const int in = pl & 1023, quad = in >> 8, wave = (in >> 6) & 3, lane = in & 63;
    i = tx * TILE + (quad & 1) * 16 + (wave & 1) * 8 + (lane & 7);
    j = ty * TILE + (quad >> 1) * 16 + (wave >> 1) * 8 + (lane >> 3);
}

// Wave-aggregated append: lanes with `want` get consecutive indices from *counter (one atomic per wave).
__device__ __forceinline__ unsigned wave_append(bool want, unsigned *counter) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const unsigned lane = threadIdx.x & 63;
    const unsigned prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
    const int leader = __ffsll((long long)mask) - 1;
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (unsigned)__popcll(mask));
    base = __shfl(base, leader, 64);
    return base + prefix;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Float atomics run at the memory side in 64-byte requests: with one plane per colour, lanes whose slots are neighbours
// share a request.
__device__ __forceinline__ void accum_add(const Wavefront &W, int slot, V v) {
    // (a contribution of zero -- a path that leaves a scene with a black background, a vertex in shadow without an ambient
    // term -- changes no sum: the sums start at +0 and x + 0 = x; skipping it saves three atomics)
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f) return;
    unsafeAtomicAdd(W.accum + (size_t)slot, v.x);
    unsafeAtomicAdd(W.accum + (size_t)W.capacity + (size_t)slot, v.y);
    unsafeAtomicAdd(W.accum + 2 * (size_t)W.capacity + (size_t)slot, v.z);
}

// The frame's ray and work counters: STAT_ROWS rows of STAT_WORDS 64-bit words, one row per 128-byte line; a
// workgroup adds to row blockIdx.x % STAT_ROWS and the host sums the rows.  (All waves of a kernel adding to one
// address would queue up behind each other at the L2.)
constexpr int STAT_ROWS = 64, STAT_WORDS = 16, STAT_USED = 14;   // words 10..13: the any-hit share of words 3..6
__device__ __forceinline__ void flush_counters(const Counters &c, unsigned long long *stats, bool detailed) {
    stats += (blockIdx.x % STAT_ROWS) * STAT_WORDS;
    uint32_t vals[10] = {c.primary, c.shadow, c.secondary, c.nodes, c.tri, c.sph, c.pln, c.shaded, c.texels, c.traced};
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        if (!detailed && (k >= 3 && k <= 6)) continue;
        uint32_t v = wave_sum(vals[k]);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(stats + k, (unsigned long long)v);
    }
}

#ifndef RT_SHARD_TURN
#define RT_SHARD_TURN 37
#endif
// The queue shard of workgroup `group` (one sample of a quarter tile) of sample `sample`: see generate_kernel.
__host__ __device__ __forceinline__ int queue_shard_of(unsigned group, unsigned sample) { return (int)((group + sample * RT_SHARD_TURN) % N_SHARDS); }

// ---- generate: camera::get_samples (src/camera.cpp:117-224) for every (pixel, sample) of the batch ----------
__global__ void __launch_bounds__(256) generate_kernel(rt_camera_desc cam, Batch b, Wavefront W, unsigned long long *__restrict__ stats) {
    const unsigned g = blockIdx.x * 256u + threadIdx.x;
    const unsigned total = (unsigned)b.n_pixels * (unsigned)b.n_samples;
    Counters c;
    memset(&c, 0, sizeof c);
    bool valid = false;
    int i = 0, j = 0, s = 0;
    if (g < total) {
        s = (int)(g / (unsigned)b.n_pixels);
        batch_pixel(b, (int)(g % (unsigned)b.n_pixels), i, j);
        valid = i < cam.resx && j < cam.resy;
        W.accum[(size_t)g] = 0.f;
        W.accum[(size_t)W.capacity + g] = 0.f;
        W.accum[2 * (size_t)W.capacity + g] = 0.f;
    }
    // A workgroup holds one sample of a quarter tile.  Its shard follows the quarter tile and turns with the sample
    // (37 is odd: 64 samples visit 64 shards), so the samples of an expensive part of the image are spread over all shards
    // whatever the batch size: a shard is walked by its own workgroups only, and the slowest shard ends a launch.  (With
    // blockIdx % 64 a batch of 408 tiles sent every quarter tile to two shards, one of 736 tiles to one.)  Per sample a
    // shard still gets its even share of workgroups, give or take one.
    const unsigned per_sample = (unsigned)b.n_pixels >> 8;   // workgroups per sample (n_pixels is a multiple of 1024)
    const int shard = queue_shard_of(blockIdx.x % per_sample, blockIdx.x / per_sample);
    const unsigned local = wave_append(valid, counter_of(W, Q_PATH, 0, shard));
    const unsigned idx = shard * W.shard_entries + local;
    if (valid && local >= W.shard_entries) {
        *W.overflow = 1u;
        valid = false;
    }
    if (valid) {
        const uint32_t pixel_index = (uint32_t)j * (uint32_t)cam.resx + (uint32_t)i;
        const uint32_t skey = sample_key(cam.seed, pixel_index, (uint32_t)s);
        V o, d;
        float time;
        camera_sample(cam, i, j, s, b.grid_n, skey, o, d, time);
        ++c.primary;
        const PathQueue &Q = W.q[0];
        Q.o_time[idx] = make_float4(o.x, o.y, o.z, b.still ? 0.0f : time);
        Q.d_key[idx] = make_float4(d.x, d.y, d.z, __uint_as_float(child_key(skey, BRANCH_ROOT)));
        Q.thr_slot[idx] = make_float4(1.f, 1.f, 1.f, __uint_as_float(g));
        Q.depth_flags[idx] = 0;
    }
    flush_counters(c, stats, false);
}

// ---- trace: bvh::intersect (src/bvh.cpp:383-480), closest hit for the paths of generation `gen` (ANY = false) or
// any hit for its shadow records (ANY = true) -------------------------------------------------------------------
// Persistent waves: the grid is sized to what the chip holds, and every wave keeps pulling rays from its shard's
// queue until the queue is empty.  Each lane walks its own ray as a small state machine (see rt_device.h); per
// iteration the wave votes on what to run:
//   refill   when at least REFILL_MIN lanes have no ray, they claim the next queue entries with one atomic
//   pop      lanes whose node is finished take the next one from their stack (or finish the ray)
//   node     lanes at an interior node test its child boxes
//   leaf     lanes at a leaf test one primitive each -- once LEAF_MIN lanes wait, or more lanes wait than walk
// Incoherent rays leave a wave's lanes at different places in the tree; running the long leaf code for one lane
// while 63 wait is what the vote avoids.  The node step runs in every iteration that has a lane for it, and the leaf
// step follows in the same iteration when its vote passes (lanes that reached a leaf in that node step
// included): no lane sits out an iteration because the other kind of step was chosen (closest-hit time -2 %).
// -DRT_SEPARATE_STEPS: one kind of step per iteration.  Each lane still executes exactly the sequential
// algorithm for its ray, so results and counters do not depend on the schedule.
struct TraceTuning {
    int refill_min;   // idle lanes that trigger a refill
    int leaf_min;     // lanes waiting at a leaf that trigger a leaf step
    int pop_min;      // lanes whose next node must come from the stack loop (not taken inside a step) that trigger that loop
};

// A finished ray's record: visibility of the light sample (any hit) or the closest hit.
template <bool ANY>
__device__ __forceinline__ void ray_result(const Wavefront &W, int gen, unsigned idx, float t, int prim, float dist, float eps) {
    if (ANY) {
        const bool in_shadow = prim != -1 && t < dist - eps;   // src/ray_trace.cpp:232-241
        W.shadow[gen & 1].vis[idx] = in_shadow ? 0.f : 1.f;
    } else {
        W.hit[idx] = make_float2(t, __int_as_float(prim));
    }
}

// WIDE: interior nodes are taken from the wide tree (four quantised boxes per 64-byte record, see wide_step) instead
// of the two-box records.  The counting variants always walk the two-box records, whose visits are the reference's.
// The any-hit walk of the wide tree gains from eight waves per SIMD where the closest-hit walk does best with six (4K /
// 64 spp frame, closest / any hit: 6 waves 230.4 / 105.1 ms, 7 waves 232.3 / 103.4, 8 waves 236.1 / 101.9).  Its stack
// entries are 4 bytes (wide_any_step), so the LDS share of eight workgroups per CU holds more levels of it than the closest-hit
// walk's eleven -- which buys nothing (any hit with 8 / 12 / 15 levels: 93.6 / 93.8 / 92.8 ms); what is left of the share goes to
// the top of the tree: 127 records instead of 63 (92.4 -> 91.4 ms; 191: 92.0; 255 records and 4 levels: 99.5).
#ifndef RT_ANY_WAVES
#define RT_ANY_WAVES 8
#endif
#ifndef RT_ANY_LDS_STACK
#define RT_ANY_LDS_STACK 8
#endif
#ifndef RT_ANY_TOP_NODES
#define RT_ANY_TOP_NODES 127
#endif
// FREE (any hit, wide): the scene's shadow rays get the same answer whatever the order of the walk (see
// shadow_answers_order_free), so they take wide_any_step: slots in record order, 4-byte stack entries.
template <bool ANY, bool COUNT, bool WIDE, bool FREE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ANY && WIDE ? RT_ANY_WAVES : RT_TRACE_WAVES, ANY && WIDE ? RT_ANY_WAVES : RT_TRACE_WAVES))) trace_kernel(DevScene S, Wavefront W, int gen, int whitted, TraceTuning tune, unsigned long long *__restrict__ stats) {
    static_assert(!(WIDE && COUNT), "the counters follow the two-box walk");
    static_assert(!FREE || (ANY && WIDE), "only any-hit rays may walk in any order");
    constexpr bool AW = FREE;   // the order-free any-hit walk of the wide tree: 4-byte stack entries
    constexpr int LEVELS = ANY && WIDE ? RT_ANY_LDS_STACK : LDS_STACK;   // stack entries in LDS
    __shared__ uint2 stack_lds[AW ? (LEVELS + 1) / 2 * 256 : LEVELS * 256];
    constexpr int TOP = AW ? RT_ANY_TOP_NODES : TOP_NODES;   // records of the tree's top in LDS (4-byte entries leave room for more)
    __shared__ float4 top_nodes[TOP > 0 ? 4 * TOP : 1];
    uint2 *lds = stack_lds + threadIdx.x;
    unsigned *lds_any = (unsigned *)stack_lds + threadIdx.x;
    uint2 spill[AW ? (64 - LEVELS + 1) / 2 : 64 - LEVELS];
    unsigned *spill_any = (unsigned *)spill;
    const int n_top = TOP > 0 ? min(WIDE ? S.n_wtop : S.n_top, TOP) : 0;
    {
        const float4 *records = WIDE ? S.wnodes : S.nodes;
        for (int i = threadIdx.x; i < 4 * n_top; i += 256) top_nodes[i] = records[i];
    }
    __syncthreads();
    const int shard = blockIdx.x % N_SHARDS;
    // any hit, `whitted` = 1: the shading kernel wrote one shadow ray per vertex, ray i for vertex i, and counted the vertices
    unsigned *ctr = counter_of(W, ANY ? (whitted ? Q_VERTEX : Q_SHADOW) : Q_PATH, gen, shard);
    const unsigned room = ANY ? (whitted ? min(W.shard_entries, W.shadow_shard_entries) : W.shadow_shard_entries) : W.shard_entries;
    unsigned count = ctr[0];
    if (count > room) count = room;
    // next unclaimed entry of this shard; shares the append counter's line, zeroed with it (word 2 of a shadow line
    // counts the light-sample records)
    unsigned *fetch = ctr + 1;
    const unsigned base_idx = shard * (ANY ? W.shadow_shard_entries : W.shard_entries);
    const PathQueue &Q = W.q[gen & 1];
    const float eps = ANY ? S.seps : S.ieps;
    const unsigned lane = threadIdx.x & 63;
    Counters c;
    memset(&c, 0, sizeof c);
    RayState r;
    r.cur = REF_IDLE;
    r.sp = 0;
    unsigned idx = 0;
    float dist = 0.f;   // any hit: distance to the light sample
    bool exhausted = count == 0;
    bool fresh = false;   // WIDE: the leaf at r.cur has just been reached, its own box is still to be tested
#ifdef RT_TRACE_PROFILE
    // diagnostic build: per wave, iterations and participating lanes of each kind of step (closest-hit kernel only),
    // reported through the work-counter words that are otherwise zero without counters
    unsigned pf_node_it = 0, pf_node_ln = 0, pf_leaf_it = 0, pf_leaf_ln = 0, pf_pop_it = 0, pf_pop_ln = 0, pf_it = 0, pf_busy = 0;
    // 3 (closest hit) / 4 (any hit): a wave's lifetime and the part of it after its shard's queue ran dry, in 100 MHz ticks
    const unsigned long long pf_t0 = wall_clock64();
    unsigned long long pf_dry = 0;
#endif

    for (;;) {
#ifdef RT_TRACE_PROFILE
        if (exhausted && !pf_dry) pf_dry = wall_clock64();
#endif
        const unsigned long long idle = __ballot(r.cur == REF_IDLE);
        if (idle) {
            if (!exhausted) {
                const int n_idle = __popcll(idle);
                if (n_idle >= tune.refill_min) {
                    const int leader = __ffsll((long long)idle) - 1;
                    unsigned first = 0;
                    if ((int)lane == leader) first = atomicAdd(fetch, (unsigned)n_idle);
                    first = __shfl(first, leader, 64);
                    if (first + (unsigned)n_idle >= count) exhausted = true;
                    const unsigned mine = first + __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0));
                    if (r.cur == REF_IDLE && mine < count) {
                        idx = base_idx + mine;
                        bool go;
                        if (ANY) {
                            const float4 od = W.shadow[gen & 1].o_dist[idx], dt = W.shadow[gen & 1].d_time[idx];
                            r.ox = od.x, r.oy = od.y, r.oz = od.z;
                            r.dx = dt.x, r.dy = dt.y, r.dz = dt.z;
                            r.time = dt.w;
                            r.culling = false;
                            dist = od.w;
                            go = ray_begin<true, COUNT>(S, r, eps, dist, c);
                        } else {
                            const float4 ot = Q.o_time[idx], dk = Q.d_key[idx];
                            const int df = Q.depth_flags[idx];
                            r.ox = ot.x, r.oy = ot.y, r.oz = ot.z;
                            r.dx = dk.x, r.dy = dk.y, r.dz = dk.z;
                            r.time = ot.w;
                            r.culling = ((df >> 16) & 1) != 0;
                            if (whitted && (float)(df & 0xffff) >= S.max_depth) {   // depth limit: no traversal, background (src/ray_trace.cpp:649-679)
                                r.t = 1e30f;
                                r.prim = -3;
                                go = false;
                            } else {
                                go = ray_begin<false, COUNT>(S, r, eps, 1e30f, c);
                            }
                        }
                        if (WIDE && go) r.cur = S.wroot_ref;
                        if (!go) {
                            ray_result<ANY>(W, gen, idx, r.t, r.prim, dist, eps);
                            r.cur = REF_IDLE;
                        }
                    }
                }
            } else if (idle == ~0ull) {
                break;
            }
        }
#ifdef RT_TRACE_PROFILE
        ++pf_it;
        pf_busy += __popcll(__ballot(r.cur != REF_IDLE));
        if (const unsigned long long m = __ballot(r.cur == REF_POP)) ++pf_pop_it, pf_pop_ln += __popcll(m);
#endif
        // (closest hit: the stack loop runs for a few lanes at a time -- entries that no longer qualify are dropped one per
        // step, spilled ones are only taken here; with pop_min > 1 it waits until that many lanes want it, or until no lane
        // has a node or a leaf to go on with.  An any-hit ray takes every entry inside its step, and only pays for the vote:
        // +6 % on a frame of short shadow rays)
        bool run_pop = true;
        if (!COUNT && !ANY && tune.pop_min > 1) {
            const int n_pop = __popcll(__ballot(r.cur == REF_POP));
            run_pop = n_pop >= tune.pop_min || __ballot(r.cur != REF_POP && r.cur != REF_IDLE) == 0;
        }
        if (run_pop && r.cur == REF_POP) {
            if (!(AW ? pop_any<LEVELS>(r, lds_any, spill_any) : stack_pop<(!ANY || COUNT), COUNT, LEVELS>(r, lds, spill, c))) {
                ray_result<ANY>(W, gen, idx, r.t, r.prim, dist, eps);
                r.cur = REF_IDLE;
            }
            fresh = true;
        }
        const unsigned long long at_node = __ballot((unsigned)r.cur < (unsigned)REF_POP);
        const unsigned long long at_leaf = __ballot(r.cur < 0);
        const int n_node = __popcll(at_node), n_leaf = __popcll(at_leaf);
#ifndef RT_SEPARATE_STEPS
        const bool run_node = n_node != 0;
#else
        const bool run_node = n_node && !(n_leaf >= tune.leaf_min || n_leaf >= n_node);
#endif
        if (run_node) {
            if ((unsigned)r.cur < (unsigned)REF_POP) {
                if (WIDE) {
                    if (AW)
                        wide_any_step<LEVELS>(S, r, lds_any, spill_any, top_nodes, n_top);
                    else
                        wide_step<LEVELS>(S, r, lds, spill, top_nodes, n_top);
#ifndef RT_NO_INLINE_POP
                    if (r.cur == REF_POP && (AW ? pop_any_inline<LEVELS>(r, lds_any) : pop_inline<!ANY, LEVELS>(r, lds))) {
                        ray_result<ANY>(W, gen, idx, r.t, r.prim, dist, eps);
                        r.cur = REF_IDLE;
                    }
#endif
                    fresh = true;
                } else
                    node_step<COUNT>(S, r, lds, spill, top_nodes, n_top, c);
            }
        }
#ifndef RT_SEPARATE_STEPS
        // after the node step: lanes that have just reached a leaf count, and take part
        const int n_leaf_now = __popcll(__ballot(r.cur < 0)), n_node_now = __popcll(__ballot((unsigned)r.cur < (unsigned)REF_POP));
        const bool run_leaf = n_leaf_now && (n_leaf_now >= tune.leaf_min || n_leaf_now >= n_node_now);
#else
        const bool run_leaf = !run_node && n_leaf;
#endif
#ifdef RT_TRACE_PROFILE
        if (run_node) ++pf_node_it, pf_node_ln += n_node;
        if (run_leaf) ++pf_leaf_it, pf_leaf_ln += __popcll(__ballot(r.cur < 0));
#endif
        if (run_leaf) {
            if (r.cur < 0) {
                const int k = r.cur & LEAF_ENTRY_MASK;
                const float4 *entry = S.prims + PRIM_STRIDE * (size_t)k;   // one 64-byte block, read at once
                const float4 p0 = entry[0], p1 = entry[1], p2 = entry[2];
                bool inside = true;
                if (WIDE && fresh) {
                    // The quantised boxes decide what the walk looks at; whether a leaf's entries are tested is decided by
                    // the leaf's own box, as in the two-box walk.  (A ray that passes the leaf's box passes the boxes of
                    // all its ancestors, which contain it, so this one test stands for the whole chain.)  The box sits in
                    // the entry's own 64-byte block.
                    fresh = false;
                    const float4 p3 = entry[3];
                    float tm;
                    inside = slab(p2.z, p2.w, p3.x, p3.y, p3.z, p3.w, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, r.ix < 0, r.iy < 0, r.iz < 0, tm) && tm < r.t;
                    if (!inside) r.cur = REF_POP;
                }
                if (inside) {
                    float tc;
                    bool done = false;
                    if (prim_test<COUNT>(S, p0, p1, p2, r, eps, tc, c) && tc > eps && tc < r.t) {
                        r.t = tc;
                        r.prim = k;
                        done = ANY;
                    }
                    if (done) {
                        ray_result<ANY>(W, gen, idx, r.t, r.prim, dist, eps);
                        r.cur = REF_IDLE;
                    } else {
                        r.cur = (__float_as_int(p2.y) & PRIM_LAST_BIT) ? REF_POP : r.cur + 1;
                    }
                }
#ifndef RT_NO_INLINE_POP
                if (WIDE && r.cur == REF_POP) {
                    if (AW ? pop_any_inline<LEVELS>(r, lds_any) : pop_inline<!ANY, LEVELS>(r, lds)) {
                        ray_result<ANY>(W, gen, idx, r.t, r.prim, dist, eps);
                        r.cur = REF_IDLE;
                    }
                    fresh = true;
                }
#endif
            }
        }
    }
#ifdef RT_TRACE_PROFILE
    if (RT_TRACE_PROFILE >= 3 && ANY == (RT_TRACE_PROFILE == 4) && lane == 0) {
        const unsigned long long t1 = wall_clock64();
        unsigned long long *row = stats + (blockIdx.x % STAT_ROWS) * STAT_WORDS;
        atomicAdd(row + 3, t1 - pf_t0);                       // lifetime
        atomicAdd(row + 4, pf_dry ? t1 - pf_dry : 0ull);      // after the queue ran dry
        atomicAdd(row + 5, 1ull);                             // waves
        atomicAdd(row + 6, (unsigned long long)pf_it);        // iterations
    }
    if (RT_TRACE_PROFILE < 3 && ANY == (RT_TRACE_PROFILE == 2) && lane == 0) {   // 1: the closest-hit kernel's waves, 2: the any-hit kernel's
        unsigned long long *row = stats + (blockIdx.x % STAT_ROWS) * STAT_WORDS;
        const unsigned vals[8] = {pf_node_it, pf_node_ln, pf_leaf_it, pf_leaf_ln, pf_pop_it, pf_pop_ln, pf_it, pf_busy};
        for (int k = 0; k < 4; ++k) atomicAdd(row + 3 + k, (unsigned long long)vals[k]);
        for (int k = 0; k < 4; ++k) atomicAdd(row + 10 + k, (unsigned long long)vals[4 + k]);
    }
#endif
    if (COUNT) {
        flush_counters(c, stats, true);
        if (ANY) {   // the same work again under the any-hit words
            unsigned long long *row = stats + (blockIdx.x % STAT_ROWS) * STAT_WORDS;
            const uint32_t vals[4] = {c.nodes, c.tri, c.sph, c.pln};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t v = wave_sum(vals[k]);
                if ((threadIdx.x & 63) == 0 && v) atomicAdd(row + 10 + k, (unsigned long long)v);
            }
        }
    }
}

__device__ __forceinline__ void put_path(const Wavefront &W, int gen, int shard, unsigned local, V o, V d, float time, V thr, uint32_t key,
                                         int slot, int depth_flags) {
    if (local >= W.shard_entries) {
        *W.overflow = 1u;
        return;
    }
    const unsigned idx = shard * W.shard_entries + local;
    const PathQueue &Q = W.q[gen & 1];
    Q.o_time[idx] = make_float4(o.x, o.y, o.z, time);
    Q.d_key[idx] = make_float4(d.x, d.y, d.z, __uint_as_float(key));
    Q.thr_slot[idx] = make_float4(thr.x, thr.y, thr.z, __int_as_float(slot));
    Q.depth_flags[idx] = depth_flags;
}

// ---- shade: the hit-side half of trace_rec (src/ray_trace.cpp:646-869) / path_trace (:871-1139) ------------------
// Everything except the light loop of calculate_color, which becomes a vertex record for the direct kernel.
// PLAIN: the scene has no textures and no mirror, conductor or dielectric material, so the code for them is left out
// (the same statements otherwise; what remains needs fewer registers and less room in the instruction cache).
// waves per SIMD the register allocation aims at, measured (4K / 64 spp frame).  The full shading kernel 3: 86 ms, 4: 81,
// 5: 106 (spills); the plain one, by the frame's time, 3: 419 ms, 4: 417, 5: 413, 6: 420.  Gather (two emitter samples per
// round, packed) 4: 58.5 ms, 5: 61.8, 6: 64.5, 8: 67.7
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4
#endif
#ifndef RT_PLAIN_SHADE_WAVES
#define RT_PLAIN_SHADE_WAVES 5
#endif
#ifndef RT_GATHER_WAVES
#define RT_GATHER_WAVES 4
#endif
template <bool PT, bool PLAIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PLAIN ? RT_PLAIN_SHADE_WAVES : RT_SHADE_WAVES))) shade_kernel(DevScene S, rt_camera_desc cam, Batch b, Wavefront W, int gen, int emitter_rays, unsigned long long *__restrict__ stats) {
    const int shard = blockIdx.x % N_SHARDS;
    const unsigned shard_base = shard * W.shard_entries;
    unsigned count = *counter_of(W, Q_PATH, gen, shard);
    if (count > W.shard_entries) count = W.shard_entries;
    const PathQueue &Q = W.q[gen & 1];
    unsigned *next_count = counter_of(W, Q_PATH, gen + 1, shard);
    unsigned *vcount = counter_of(W, Q_VERTEX, gen, shard);
    const bool is_hdr = cam.is_hdr != 0;
    Counters c;
    memset(&c, 0, sizeof c);
    const unsigned stride = (gridDim.x / N_SHARDS) * 256u;
    for (unsigned base = (blockIdx.x / N_SHARDS) * 256u + (threadIdx.x & ~63u); base < count; base += stride) {
        const unsigned li = base + (threadIdx.x & 63u);
        const bool active = li < count;
        const unsigned idx = shard_base + li;
        // per-lane outputs of this pass
        bool want_vertex = false;
        V v_p = mk(0, 0, 0), v_n = mk(0, 0, 0), v_view = mk(0, 0, 0), v_kd = mk(0, 0, 0), v_ks = mk(0, 0, 0), v_w = mk(0, 0, 0), v_amb = mk(0, 0, 0);
        float v_scale = 1.0f;
        int v_mat = 0;
        int n_children = 0;          // explicit children 0 (reflect) and 1 (refract), or `n_children` diffuse bounces
        bool diffuse = false;
        V c_o[2], c_d[2], c_thr[2];
        uint32_t c_key[2];
        int c_df[2];
        V b_origin = mk(0, 0, 0), b_normal = mk(0, 0, 0), b_kd = mk(0, 0, 0), b_thr = mk(0, 0, 0);
        float b_helper_x = 0.f, b_rr = 1.f;
        uint32_t key = 0;
        int slot = 0, depth = 0;
        bool culling = false;
        float time = 0.f;
        if (active) {
            const float4 ot = Q.o_time[idx], dk = Q.d_key[idx], ts = Q.thr_slot[idx];
            const int df = Q.depth_flags[idx];
            const V o = mk(ot.x, ot.y, ot.z), d = mk(dk.x, dk.y, dk.z), thr = mk(ts.x, ts.y, ts.z);
            time = ot.w;
            key = __float_as_uint(dk.w);
            slot = __float_as_int(ts.w);
            depth = df & 0xffff;
            culling = ((df >> 16) & 1) != 0;
            const bool diffuse_bounce = ((df >> 17) & 1) != 0;
            Hit h;
            const float2 found = W.hit[idx];
            h.prim = __float_as_int(found.y);
            h.t = h.prim == -3 ? 0.f : found.x;
            float pixelu = 0.f, pixelv = 0.f;
            if (!PT && (h.prim == -1 || h.prim == -3) && !is_hdr && S.bg_tex >= 0) {
                // LDR background images are looked up by pixel position (src/main.cpp:455-456); slots are laid out
                // sample-major over the batch's pixel-local indices
                int pi, pj;
                batch_pixel(b, slot % b.n_pixels, pi, pj);
                pixelv = 1.0f - rt_div((float)pj, (float)(cam.resy - 1));
                pixelu = rt_div((float)pi, (float)(cam.resx - 1));
            }
            if (h.prim == -1 || h.prim == -3) {
                V bgc;
                if (PT && diffuse_bounce && S.bg_tex >= 0 && S.textures[S.bg_tex].image >= 0) {
                    // a diffuse-bounce miss under an environment map fetches one fixed texel (the reference reads
                    // zeroed scalar copies of the sampled direction, src/ray_trace.cpp:10-12,1093-1116)
                    bgc = mk(0, 0, 0);
                    tex_sample(S, S.textures[S.bg_tex], S.is_probe ? 0.5f : 1.0f, 0.5f, mk(0, 0, 0), bgc, c);
                } else {
                    bgc = background(S, cam, d, !PT, pixelu, pixelv, c);
                }
                accum_add(W, slot, mul(thr, bgc));
            } else {
                ++c.shaded;
                const Surface sf = surface_of(S, h);
                const rt_material &m = S.materials[sf.material];
                V p = add(o, scl(d, h.t));
                V normal = normal_of(S, sf, p);
                const float helper_x = sf.kind == 2 ? S.planes[sf.index].raw_normal_x : normal.x;
                if (!PLAIN) {
                    apply_normal_map(S, sf, p, normal, c);
                    apply_bump_map(S, sf, p, normal, c);
                }
                const float seps = S.seps;
                const int child_df = (depth + 1) | ((int)culling << 16);
                if (!PT) {
                    // ---- trace_rec ---------------------------------------------------------------------
                    V shade_p = p, shade_n = normal;
                    if (PLAIN) {
                    } else if (m.type == RT_MAT_MIRROR || m.type == RT_MAT_CONDUCTOR) {
                        V R = reflect_dir(normal, d, m.roughness, key);
                        V coef = mk(m.mirror);
                        if (m.type == RT_MAT_CONDUCTOR) {
                            float Fs = conductor_fresnel(absf(dot(d, normal)), m.refraction_index, m.absorption_index);
                            coef = mul(mk(Fs, Fs, Fs), mk(m.mirror));
                        }
                        p = add(p, scl(normal, seps));   // the reference offsets the hit point in place (:736-743)
                        shade_p = p;
                        n_children = 1;
                        c_o[0] = p;
                        c_d[0] = R;
                        c_thr[0] = mul(thr, coef);
                        c_key[0] = child_key(key, BRANCH_REFLECT);
                        c_df[0] = child_df;
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
                        shade_n = normal;
                        float cos_i = -cos_theta;
                        cos_i = cos_i < 0.0f ? 0.0f : (cos_i > 1.0f ? 1.0f : cos_i);
                        float r0 = rt_div(n1 - n2, n1 + n2);
                        r0 = r0 * r0;
                        const float F = r0 + (1.0f - r0) * rt_pow(1.0f - cos_i, 5.0f);
                        n_children = 1;
                        c_o[0] = add(p, scl(normal, seps));
                        c_d[0] = reflect_dir(normal, d, m.roughness, key);
                        c_thr[0] = scl(mul(thr, mk(m.mirror)), F);
                        c_key[0] = child_key(key, BRANCH_REFLECT);
                        c_df[0] = child_df;
                        V tdir;
                        if (refract_dir(normal, d, n1, n2, m.roughness, key, tdir)) {
                            V att = mk(1.f, 1.f, 1.f);
                            if (!entering) {
                                V a = scl(mk(m.absorption_coefficient), -h.t);
                                att = mk(rt_exp(a.x), rt_exp(a.y), rt_exp(a.z));
                            }
                            n_children = 2;
                            c_o[1] = add(p, scl(normal, -seps));
                            c_d[1] = tdir;
                            c_thr[1] = scl(mul(thr, att), 1.0f - F);
                            c_key[1] = child_key(key, BRANCH_REFRACT);
                            c_df[1] = depth + 1;   // refracted rays never cull (:845)
                        }
                    }
                    // own colour: calculate_color(include_ambient = true) at (shade_p, shade_n)
                    V kd = mk(m.diffuse), ks = mk(m.specular), kd_rr = kd, rep = mk(0, 0, 0);
                    bool replace_all = false;
                    if (!PLAIN) resolve_textures(S, sf, shade_p, m, kd, ks, kd_rr, replace_all, rep, c);
                    if (replace_all) {
                        accum_add(W, slot, mul(thr, rep));
                    } else {
                        want_vertex = true;   // ambient + lights are summed by the gather kernel in calculate_color's order
                        v_p = shade_p;
                        v_n = shade_n;
                        v_view = normalize(sub(o, shade_p));
                        v_kd = kd;
                        v_ks = ks;
                        v_w = thr;
                        v_scale = is_hdr ? 1.0f : 0.004f;
                        v_amb = mul(mk(m.ambient), mk(S.ambient));
                        v_mat = sf.material;
                    }
                } else {
                    // ---- path_trace ------------------------------------------------------------------------
                    V kd = mk(m.diffuse), ks = mk(m.specular), kd_rr = kd, rep = mk(0, 0, 0);
                    bool replace_all = false;
                    const uint32_t texels_before = c.texels;
                    if (!PLAIN) resolve_textures(S, sf, p, m, kd, ks, kd_rr, replace_all, rep, c);
                    const uint32_t texels_resolve = c.texels - texels_before;
                    if (replace_all) {
                        accum_add(W, slot, mul(thr, rep));
                    } else {
                        V emission = sf.kind == 0 ? mk(S.instances[sf.instance].emission)
                                                  : (sf.kind == 1 ? mk(S.spheres[sf.index].emission) : mk(0, 0, 0));
                        if (!PLAIN && m.type != RT_MAT_REGULAR) {
                            if (!((float)depth < S.max_depth)) {
                                accum_add(W, slot, mul(thr, emission));   // specular surface at the depth limit keeps its emission only
                            } else if (m.type == RT_MAT_MIRROR || m.type == RT_MAT_CONDUCTOR) {
                                V coef = mk(m.mirror);
                                if (m.type == RT_MAT_CONDUCTOR) {
                                    float Fs = conductor_fresnel(absf(dot(d, normal)), m.refraction_index, m.absorption_index);
                                    coef = mul(mk(Fs, Fs, Fs), mk(m.mirror));
                                }
                                n_children = 1;
                                c_o[0] = add(p, scl(normal, seps));
                                c_d[0] = reflect_dir(normal, d, m.roughness, key);
                                c_thr[0] = mul(thr, coef);
                                c_key[0] = child_key(key, BRANCH_REFLECT);
                                c_df[0] = child_df;
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
                                V T;
                                const bool refracts = refract_dir(rn, d, n1, n2, m.roughness, key, T);
                                n_children = 1;
                                c_o[0] = add(p, scl(rn, seps));
                                c_d[0] = reflect_dir(rn, d, m.roughness, key);
                                c_thr[0] = refracts ? scl(mul(thr, mk(m.mirror)), F) : mul(thr, mk(m.mirror));
                                c_key[0] = child_key(key, BRANCH_REFLECT);
                                c_df[0] = child_df;
                                if (refracts) {
                                    V att = mk(1.f, 1.f, 1.f);
                                    if (!entering) {
                                        V a = scl(mk(m.absorption_coefficient), -h.t);
                                        att = mk(rt_exp(a.x), rt_exp(a.y), rt_exp(a.z));
                                    }
                                    n_children = 2;
                                    c_o[1] = add(p, scl(rn, -seps));
                                    c_d[1] = T;
                                    c_thr[1] = scl(mul(thr, att), 1.0f - F);
                                    c_key[1] = child_key(key, BRANCH_REFRACT);
                                    c_df[1] = child_df;
                                }
                            }
                        } else {
                            accum_add(W, slot, mul(thr, emission));
                            float rr_prob = 1.0f;
                            bool alive = true;
                            if (cam.russian_roulette && depth >= cam.min_recursion_depth) {
                                float max_refl = max_std(max_std(kd_rr.x, kd_rr.y), kd_rr.z);
                                rr_prob = min_std(max_refl, 0.99f);
                                if (rnd(key, DIM_RR) > rr_prob) alive = false;
                            }
                            if (alive) {
                                if (cam.next_event_estimation) {
                                    c.texels += texels_resolve;   // the reference resolves the textures again inside calculate_color
                                    if (S.n_lights > 0) {
                                        want_vertex = true;
                                        v_p = p;
                                        v_n = normal;
                                        v_view = normalize(sub(o, p));
                                        v_kd = kd;
                                        v_ks = ks;
                                        v_w = thr;
                                        v_scale = 0.004f;   // path_trace calls calculate_color with is_hdr = false (:1083)
                                        v_mat = sf.material;
                                    }
                                }
                                if ((float)depth < S.max_depth) {
                                    diffuse = true;
                                    n_children = (depth == 0 && cam.splitting_factor > 1) ? cam.splitting_factor : 1;
                                    b_origin = add(p, scl(normal, seps));
                                    b_normal = normal;
                                    b_helper_x = helper_x;
                                    b_kd = kd;
                                    b_thr = thr;
                                    b_rr = rr_prob;
                                }
                            }
                        }
                    }
                }
            }
        }
        // ---- append the vertex record ------------------------------------------------------------------------
        {
            const unsigned vl = wave_append(want_vertex, vcount);
            if (want_vertex) {
                if (vl >= W.shard_entries) {
                    *W.overflow = 1u;
                } else {
                    const unsigned vi = shard_base + vl;
                    const rt_material &vm = S.materials[v_mat];
                    const int slot_scale = slot | (v_scale != 1.0f ? (int)0x80000000 : 0);
                    V ldir = mk(0, 0, 0);
                    float dist = 0.f, d2 = 1.f;
                    if (emitter_rays) sphere_light_geometry(S.spheres[S.lights[0].sphere], v_p, ldir, dist, d2);
                    if (PT && emitter_rays) {
                        // Path tracing under one sphere emitter (gather mode 3): no ambient term, no light-sample records, and the
                        // direction to the emitter does not depend on the sample -- so what the gather needs of the vertex is the
                        // BRDF's two factors for that direction, the squared distance, the path's weight, key and slot: 48 bytes
                        // instead of 96 (the direction itself is in the shadow ray below).  Same functions on the same inputs as the
                        // gather would call: the same bits.
                        V f_d, f_s;
                        brdf_terms(brdf_code(vm), vm.phong_exponent, v_n, ldir, v_view, v_kd, v_ks, f_d, f_s);
                        W.vtx[gen & 1].n_key[vi] = make_float4(f_d.x, f_d.y, f_d.z, __uint_as_float(key));
                        W.vtx[gen & 1].view_mat[vi] = make_float4(f_s.x, f_s.y, f_s.z, d2);
                        W.vtx[gen & 1].kd_slot[vi] = make_float4(v_w.x, v_w.y, v_w.z, __int_as_float(slot_scale));
                    } else {
                        W.vtx[gen & 1].p_time[vi] = make_float4(v_p.x, v_p.y, v_p.z, time);
                        W.vtx[gen & 1].n_key[vi] = make_float4(v_n.x, v_n.y, v_n.z, __uint_as_float(key));
                        W.vtx[gen & 1].view_mat[vi] = make_float4(v_view.x, v_view.y, v_view.z, __int_as_float(brdf_code(vm)));
                        W.vtx[gen & 1].kd_slot[vi] = make_float4(v_kd.x, v_kd.y, v_kd.z, __int_as_float(slot_scale));
                        W.vtx[gen & 1].ks_n[vi] = make_float4(v_ks.x, v_ks.y, v_ks.z, vm.phong_exponent);
                        W.vtx[gen & 1].weight[vi] = make_float4(v_w.x, v_w.y, v_w.z, 0.f);
                        W.vtx[gen & 1].amb_first[vi] = make_float4(v_amb.x, v_amb.y, v_amb.z, 0.f);
                    }
                    if (emitter_rays) {
                        // The scene's only light is one sphere emitter, whose samples share one shadow ray per vertex (see
                        // light_setup_kernel): the ray is written here, ray i for vertex i, and no set-up pass runs.
                        const V so = add(v_p, scl(ldir, S.seps));
                        const unsigned ri = shard * W.shadow_shard_entries + vl;
                        if (vl < W.shadow_shard_entries) {
                            ++c.traced;
                            W.shadow[gen & 1].o_dist[ri] = make_float4(so.x, so.y, so.z, dist);
                            W.shadow[gen & 1].d_time[ri] = make_float4(ldir.x, ldir.y, ldir.z, time);
                        } else {
                            *W.overflow = 1u;
                        }
                    }
                }
            }
        }
        // ---- append child paths, one ballot round per child index -------------------------------------------------
        int max_children = n_children;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) max_children = max(max_children, __shfl_xor(max_children, off, 64));
        for (int r = 0; r < max_children; ++r) {
            const bool want = r < n_children;
            const unsigned ci = wave_append(want, next_count);
            if (!want) continue;
            ++c.secondary;
            if (!PLAIN && !diffuse) {
                put_path(W, gen + 1, shard, ci, c_o[r], c_d[r], time, c_thr[r], c_key[r], slot, c_df[r]);
            } else {
                const bool cosine = cam.importance_sampling != 0;
                V wi;
                float pdf;
                if (n_children > 1)
                    sample_hemisphere(cosine, rnd(key, DIM_SPLIT + 2 * r), rnd(key, DIM_SPLIT + 2 * r + 1), b_normal, b_helper_x, wi, pdf);
                else
                    sample_hemisphere(cosine, rnd(key, DIM_HEMI_1), rnd(key, DIM_HEMI_2), b_normal, b_helper_x, wi, pdf);
                float ndotl = max_std(dot(b_normal, wi), 0.0f);
                V w = mk(0, 0, 0);
                if (pdf > 1e-6f) w = scl(scl(b_kd, 1.0f / 3.14159265359f), rt_div(ndotl, pdf));
                w = scl(w, rt_div(1.0f, (float)n_children));
                if (b_rr < 1.0f && b_rr > 0.0f) w = scl(w, rt_div(1.0f, b_rr));
                put_path(W, gen + 1, shard, ci, b_origin, wi, time, mul(b_thr, w), child_key(key, BRANCH_BOUNCE + r), slot,
                         (depth + 1) | ((int)culling << 16) | (1 << 17));
            }
        }
    }
    flush_counters(c, stats, false);
}

// ---- direct lighting = the light loop of calculate_color (src/ray_trace.cpp:182-400) in three kernels --------------
// light_setup   one lane per vertex, looping over the light samples of this pass (so the light is the same for the
//               whole wave and its data comes through scalar loads): Light::get_sample, BRDF terms, and -- when the
//               incident radiance is non-zero -- a light-sample record plus a shadow ray, appended with wave
//               ballots unless the vertex's previous record has the same ray
// trace (any)   one lane per shadow ray: any-hit traversal, visibility = !(hit && t < dist - epsilon)
// light_gather  one lane per vertex: ambient, then the visible records' diffuse and specular terms added in
//               light-sample order (the order of the reference's loop), LDR scale, throughput, add to the slot
// Sphere emitters are deferred: every sample of one aims at the sphere's centre (src/lights.cpp:418-422), so the
// emitter has ONE shadow ray per vertex.  light_setup writes only that ray; light_gather evaluates the emitter's samples
// (random numbers, emitter normal, radiance, BRDF) for the vertices the ray reaches and adds them in the same place of
// the sample order -- no records, and no sample work behind an occluder.  With the work counters on nothing is deferred
// or shared: every sample is evaluated and traced as in the reference's loop.
// Light samples are numbered over all lights (`light_of` maps a number to its light); a pass covers 64 of them.
// Records are stored per block of 64 vertices, sample-major: record (block, sample, lane), so that both kernels
// read and write 64 consecutive records at a time.
__device__ __forceinline__ unsigned record_index(unsigned shadow_base, unsigned vbase, unsigned k_pad, unsigned slot, unsigned lane) {
    return shadow_base + (vbase * k_pad + slot * 64u) + lane;   // vbase is a multiple of 64, slot < k_pad
}
// The record slot of light sample k (number kl of its pass).  With every sample recorded it is kl; with deferred emitters
// only the other samples and one entry per emitter need a slot, and a table behind `light_of` numbers those per pass, so
// that a vertex takes k_pad = a few slots instead of 16 or 64.
__device__ __forceinline__ unsigned record_slot(const int *__restrict__ light_of, int slot_table, unsigned k, int kl) {
    return slot_table >= 0 ? (unsigned)light_of[slot_table + (int)k] : (unsigned)kl;
}

// SPHERES: the pass holds samples of a sphere emitter.  A pass without any is set up by a build that leaves that code and the
// registers it keeps across an emitter's samples out.  Waves per SIMD the register allocation aims at, measured on the
// textured 1080p / 64 spp frame (two area lights of 16 samples): 3 (139 registers, the compiler's own choice): 78.4 ms,
// 4: 66.4, 5: 88.7 (spills), 6: 145; the build without sphere emitters 3: 69.0, 4: 63.0, 5: 58.5, 6: 59.3.
#ifndef RT_SETUP_WAVES
#define RT_SETUP_WAVES 4
#endif
#ifndef RT_SETUP_WAVES_NO_SPHERES
#define RT_SETUP_WAVES_NO_SPHERES 5
#endif
template <bool SPHERES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SPHERES ? RT_SETUP_WAVES : RT_SETUP_WAVES_NO_SPHERES))) light_setup_kernel(DevScene S, Wavefront W, int gen, int pass, const int *__restrict__ light_of,
                                                           int k_total, int k_pad, int slot_table, int share_rays,
                                                           unsigned long long *__restrict__ stats) {
    const int shard = blockIdx.x % N_SHARDS;
    const unsigned shard_base = shard * W.shard_entries, shadow_base = shard * W.shadow_shard_entries;
    unsigned nv = *counter_of(W, Q_VERTEX, gen, shard);
    if (nv > W.shard_entries) nv = W.shard_entries;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave_global = ((blockIdx.x / N_SHARDS) * 256u + threadIdx.x) >> 6;   // wave number within the shard
    const unsigned n_waves = ((gridDim.x / N_SHARDS) * 256u) >> 6;
    const int k0 = pass * 64, kn = min(k_total - k0, 64);
    unsigned *ray_count = counter_of(W, Q_SHADOW, gen, shard);
    Counters c;
    memset(&c, 0, sizeof c);
    const float seps = S.seps;
    for (unsigned vbase = wave_global * 64u; vbase < nv; vbase += n_waves * 64u) {
        const unsigned vl = vbase + lane;
        const bool valid = vl < nv;
        const unsigned v = shard_base + (valid ? vl : vbase);
        const float4 pt = W.vtx[gen & 1].p_time[v];
        const V p = mk(pt.x, pt.y, pt.z);
        // the rest of the vertex is read when the first light that is evaluated here comes up (a deferred one needs none)
        float4 nk = make_float4(0, 0, 0, 0), vm = nk, ksn = nk;
        V normal = mk(0, 0, 0), view = normal, kd = normal, ks = normal;
        uint32_t key = 0;
        bool vertex_read = false;
        const bool room = record_index(shadow_base, vbase, (unsigned)k_pad, (unsigned)k_pad - 1u, 63u) < shadow_base + W.shadow_shard_entries;
        if (!room) *W.overflow = 1u;
        unsigned long long have = 0;
        // the vertex's previous record: its ray, for sharing.  Two samples have the same ray exactly when they belong
        // to the same light and that light's position does not depend on the sample.
        unsigned prev_ray = 0;
        int prev_light = -1;   // light of the vertex's previous record when it is such a light, else -1
        // a sphere light's direction, distance and BRDF factors do not depend on the sample: kept across its samples
        int kept_light = -1;
        bool kept_brdf = false;
        V kept_ldir = mk(0, 0, 0), kept_fd = mk(0, 0, 0), kept_fs = mk(0, 0, 0);
        float kept_dist = 0.f, kept_d2 = 0.f;
        for (int kl = 0; kl < kn; ++kl) {
            const unsigned k = (unsigned)(k0 + kl);
            const int li = light_of[k];
            const rt_light &L = S.lights[li];
            V rad, ldir, fd = mk(0, 0, 0), fs = mk(0, 0, 0);
            float dist;
            bool want;
            if (SPHERES && share_rays != 0 && L.type == RT_LIGHT_SPHERE) {
                // Deferred: only the light's one shadow ray per vertex is set up here, at the light's first sample of the
                // pass; the samples themselves are evaluated by light_gather for the vertices the ray reaches.
                if (li != kept_light) {   // (uniform over the wave)
                    kept_light = li;
                    sphere_light_geometry(S.spheres[L.sphere], p, kept_ldir, kept_dist, kept_d2);
                    // every vertex of the block gets a ray, so their rays are consecutive: ray of lane i = ray of lane 0 + i,
                    // and only lane 0's index is recorded (in the slot of its record for this sample)
                    const unsigned ray_local = wave_append(valid, ray_count);
                    if (valid && ray_local >= W.shadow_shard_entries) *W.overflow = 1u;
                    if (valid && room) {
                        const unsigned ray_index = shadow_base + min(ray_local, W.shadow_shard_entries - 1u);
                        ++c.traced;
                        const V so = add(p, scl(kept_ldir, seps));
                        W.shadow[gen & 1].o_dist[ray_index] = make_float4(so.x, so.y, so.z, kept_dist);
                        W.shadow[gen & 1].d_time[ray_index] = make_float4(kept_ldir.x, kept_ldir.y, kept_ldir.z, pt.w);
                        if (lane == 0)
                            ((float *)&W.shadow[gen & 1].fd_ray[record_index(shadow_base, vbase, (unsigned)k_pad, record_slot(light_of, slot_table, k, kl), 0u)])[3] = __uint_as_float(ray_index);
                    }
                }
                prev_light = -1;
                kl = min(kn, light_of[k_total + li] + L.samples - k0) - 1;   // on to the next light
                continue;
            }
            if (!vertex_read) {   // (uniform over the wave)
                vertex_read = true;
                const float4 kds = W.vtx[gen & 1].kd_slot[v];
                nk = W.vtx[gen & 1].n_key[v], vm = W.vtx[gen & 1].view_mat[v], ksn = W.vtx[gen & 1].ks_n[v];
                normal = mk(nk.x, nk.y, nk.z), view = mk(vm.x, vm.y, vm.z), kd = mk(kds.x, kds.y, kds.z), ks = mk(ksn.x, ksn.y, ksn.z);
                key = __float_as_uint(nk.w);
            }
            const float r1 = rnd(key, DIM_LIGHT + 2 * k), r2 = rnd(key, DIM_LIGHT + 2 * k + 1);
            if (SPHERES && L.type == RT_LIGHT_SPHERE) {
                const rt_sphere &sph = S.spheres[L.sphere];
                if (li != kept_light) {
                    kept_light = li;
                    kept_brdf = false;
                    sphere_light_geometry(sph, p, kept_ldir, kept_dist, kept_d2);
                }
                ldir = kept_ldir;
                dist = kept_dist;
                rad = sphere_light_radiance(L, sph, ldir, kept_d2, r1, r2);
                want = valid && !(rad.x == 0 && rad.y == 0 && rad.z == 0);
                if (want) {
                    if (!kept_brdf) {
                        kept_brdf = true;
                        brdf_terms(__float_as_int(vm.w), ksn.w, normal, ldir, view, kd, ks, kept_fd, kept_fs);
                    }
                    fd = kept_fd;
                    fs = kept_fs;
                }
            } else {
                light_sample(S, L, nullptr, p, normal, r1, r2, rad, ldir, dist, c);
                want = valid && !(rad.x == 0 && rad.y == 0 && rad.z == 0);
                if (want) brdf_terms(__float_as_int(vm.w), ksn.w, normal, ldir, view, kd, ks, fd, fs);
            }
            V so = mk(0, 0, 0);
            if (want) {
                ++c.shadow;
                so = add(p, scl(ldir, seps));
                const float lw = rt_div(1.0f, (float)L.samples);
                fd = scl(mul(fd, rad), lw);
                fs = scl(mul(fs, rad), lw);
                have |= 1ull << kl;
            }
            // a sample whose ray is the ray of the vertex's previous record shares it: same origin, direction and length,
            // computed from the same inputs by the same instructions
            const bool fixed_position = L.type == RT_LIGHT_SPHERE || L.type == RT_LIGHT_POINT || L.type == RT_LIGHT_SPOT || L.type == RT_LIGHT_DIRECTIONAL;
            const bool same = share_rays != 0 && want && fixed_position && prev_light == li;
            const bool owner = want && !same;   // this sample's ray gets traced
            const unsigned ray_local = wave_append(owner, ray_count);
            if (owner && ray_local >= W.shadow_shard_entries) *W.overflow = 1u;
            if (want && room) {
                unsigned ray_index = prev_ray;
                if (owner) {
                    ray_index = shadow_base + min(ray_local, W.shadow_shard_entries - 1u);
                    ++c.traced;
                    W.shadow[gen & 1].o_dist[ray_index] = make_float4(so.x, so.y, so.z, dist);
                    W.shadow[gen & 1].d_time[ray_index] = make_float4(ldir.x, ldir.y, ldir.z, pt.w);
                }
                const unsigned ri = record_index(shadow_base, vbase, (unsigned)k_pad, record_slot(light_of, slot_table, k, kl), lane);
                W.shadow[gen & 1].fd_ray[ri] = make_float4(fd.x, fd.y, fd.z, __uint_as_float(ray_index));
                W.shadow[gen & 1].fs_[ri] = make_float4(fs.x, fs.y, fs.z, 0.f);
                prev_ray = ray_index;
                prev_light = fixed_position ? li : -1;
            }
        }
        if (valid && (have != 0 || pass > 0)) {   // which of the vertex's samples have a record (shade_kernel wrote "none")
            ((float *)&W.vtx[gen & 1].amb_first[v])[3] = __uint_as_float((unsigned)have);
            if (kn > 32) ((float *)&W.vtx[gen & 1].weight[v])[3] = __uint_as_float((unsigned)(have >> 32));
        }
    }
    flush_counters(c, stats, false);
}

// Adds the direct lighting of one vertex (lane-local): every light sample of the pass in sample order.  A sample of a
// deferred sphere light is evaluated here when the light's ray arrived; any other sample contributes its record.
__device__ __forceinline__ V gather_vertex(const DevScene &S, const Wavefront &W, int gen, unsigned v, unsigned vl, unsigned shadow_base,
                                           const int *__restrict__ light_of, int k_total, int k0, int kn, int k_pad, int slot_table, int defer,
                                           bool active, unsigned long long have, V c, Counters &cnt) {
    const VertexBuf &B = W.vtx[gen & 1];
    const ShadowQueue &Q = W.shadow[gen & 1];
    const unsigned vbase = vl & ~63u, rec_lane = vl & 63u;
    const float4 pt = B.p_time[v], nk = B.n_key[v];
    const V p = mk(pt.x, pt.y, pt.z);
    const uint32_t key = __float_as_uint(nk.w);
    unsigned ray = 0xffffffffu;
    float vis = 0.f;
    for (int kl = 0, kend; kl < kn; kl = kend) {   // light by light; the samples of a light are consecutive
        const int li = light_of[k0 + kl];
        const rt_light &L = S.lights[li];
        kend = min(kn, light_of[k_total + li] + L.samples - k0);
        if (defer != 0 && L.type == RT_LIGHT_SPHERE) {
            const rt_sphere &sph = S.spheres[L.sphere];
            bool lit = false;
            if (defer >= 2) {   // the shading kernel wrote the rays, ray i for vertex i: the lanes that are here are those it reached
                lit = active;
            } else if (active) {   // the ray of the block's lane 0 is on record, the block's rays are consecutive
                const unsigned ri = record_index(shadow_base, vbase, (unsigned)k_pad, record_slot(light_of, slot_table, (unsigned)(k0 + kl), kl), 0u);
                lit = Q.vis[__float_as_uint(((const float *)&Q.fd_ray[ri])[3]) + rec_lane] != 0.f;
            }
            if (__ballot(lit) == 0) continue;
            // what every lane needs of the emitter, read once (the light is the same for the whole wave)
            const SphereEmitter E = sphere_emitter(L, sph);
            const float lw = rt_div(1.0f, (float)L.samples);
            V ldir = mk(0, 0, 0), f_d = mk(0, 0, 0), f_s = mk(0, 0, 0);
            float dist = 0.f, d2 = 1.f;
            if (lit) {
                sphere_light_geometry(sph, p, ldir, dist, d2);
                // the BRDF factors do not depend on the sample (a vertex whose samples all face away does not use them)
                const float4 vm = B.view_mat[v], kds = B.kd_slot[v], ksn = B.ks_n[v];
                brdf_terms(__float_as_int(vm.w), ksn.w, mk(nk.x, nk.y, nk.z), ldir, mk(vm.x, vm.y, vm.z), mk(kds.x, kds.y, kds.z),
                           mk(ksn.x, ksn.y, ksn.z), f_d, f_s);
            }
            // two samples per round: their arithmetic runs as two-wide packed operations, element for element that of
            // sphere_light_radiance; the sums are taken in sample order
            int ks = kl;
            for (; ks + 1 < kend; ks += 2) {
                if (!lit) continue;
                const unsigned k = (unsigned)(k0 + ks);
                const f2 r1 = f2{rnd(key, DIM_LIGHT + 2 * k), rnd(key, DIM_LIGHT + 2 * k + 2)};
                const f2 r2 = f2{rnd(key, DIM_LIGHT + 2 * k + 1), rnd(key, DIM_LIGHT + 2 * k + 3)};
                f2 rx, ry, rz;
                sphere_light_radiance2(E, ldir, d2, r1, r2, rx, ry, rz);
                const f2 dx = (f_d.x * rx) * lw, dy = (f_d.y * ry) * lw, dz = (f_d.z * rz) * lw;
                const f2 sx = (f_s.x * rx) * lw, sy = (f_s.y * ry) * lw, sz = (f_s.z * rz) * lw;
                if (!(rx.x == 0 && ry.x == 0 && rz.x == 0)) {
                    ++cnt.shadow;
                    c = add(c, mk(dx.x, dy.x, dz.x));
                    c = add(c, mk(sx.x, sy.x, sz.x));
                }
                if (!(rx.y == 0 && ry.y == 0 && rz.y == 0)) {
                    ++cnt.shadow;
                    c = add(c, mk(dx.y, dy.y, dz.y));
                    c = add(c, mk(sx.y, sy.y, sz.y));
                }
            }
            for (; ks < kend; ++ks) {   // an odd sample count's last one
                if (!lit) continue;
                const unsigned k = (unsigned)(k0 + ks);
                const float r1 = rnd(key, DIM_LIGHT + 2 * k), r2 = rnd(key, DIM_LIGHT + 2 * k + 1);
                const V rad = sphere_light_radiance(L, sph, ldir, d2, r1, r2);
                if (rad.x == 0 && rad.y == 0 && rad.z == 0) continue;
                ++cnt.shadow;
                c = add(c, scl(mul(f_d, rad), lw));
                c = add(c, scl(mul(f_s, rad), lw));
            }
            continue;
        }
        for (int ks = kl; ks < kend; ++ks) {
            if (!((have >> ks) & 1ull)) continue;
            const unsigned ri = record_index(shadow_base, vbase, (unsigned)k_pad, record_slot(light_of, slot_table, (unsigned)(k0 + ks), ks), rec_lane);
            const float4 fd = Q.fd_ray[ri];
            if (__float_as_uint(fd.w) != ray) {
                ray = __float_as_uint(fd.w);
                vis = Q.vis[ray];
            }
            if (vis == 0.f) continue;
            const float4 fs = Q.fs_[ri];
            c = add(c, mk(fd.x, fd.y, fd.z));
            c = add(c, mk(fs.x, fs.y, fs.z));
        }
    }
    return c;
}

// The evaluation for gather mode 3 (path tracing under one sphere emitter, see shade_kernel): the vertex record holds the
// BRDF's factors for the emitter's direction, the squared distance and the path's weight; the direction is the shadow ray's.
// The sample loop is gather_vertex's.
__device__ __forceinline__ V gather_lean(const DevScene &S, bool lit, V f_d, V f_s, V ldir, float d2, uint32_t key, Counters &cnt) {
    const rt_light &L = S.lights[0];
    const SphereEmitter E = sphere_emitter(L, S.spheres[L.sphere]);
    const float lw = rt_div(1.0f, (float)L.samples);
    V c = mk(0, 0, 0);
    int ks = 0;
    for (; ks + 1 < L.samples; ks += 2) {
        if (!lit) continue;
        const unsigned k = (unsigned)ks;
        const f2 r1 = f2{rnd(key, DIM_LIGHT + 2 * k), rnd(key, DIM_LIGHT + 2 * k + 2)};
        const f2 r2 = f2{rnd(key, DIM_LIGHT + 2 * k + 1), rnd(key, DIM_LIGHT + 2 * k + 3)};
        f2 rx, ry, rz;
        sphere_light_radiance2(E, ldir, d2, r1, r2, rx, ry, rz);
        const f2 dx = (f_d.x * rx) * lw, dy = (f_d.y * ry) * lw, dz = (f_d.z * rz) * lw;
        const f2 sx = (f_s.x * rx) * lw, sy = (f_s.y * ry) * lw, sz = (f_s.z * rz) * lw;
        if (!(rx.x == 0 && ry.x == 0 && rz.x == 0)) {
            ++cnt.shadow;
            c = add(c, mk(dx.x, dy.x, dz.x));
            c = add(c, mk(sx.x, sy.x, sz.x));
        }
        if (!(rx.y == 0 && ry.y == 0 && rz.y == 0)) {
            ++cnt.shadow;
            c = add(c, mk(dx.y, dy.y, dz.y));
            c = add(c, mk(sx.y, sy.y, sz.y));
        }
    }
    for (; ks < L.samples; ++ks) {   // an odd sample count's last one
        if (!lit) continue;
        const unsigned k = (unsigned)ks;
        const float r1 = rnd(key, DIM_LIGHT + 2 * k), r2 = rnd(key, DIM_LIGHT + 2 * k + 1);
        const V rad = sphere_light_radiance(L, S.spheres[L.sphere], ldir, d2, r1, r2);
        if (rad.x == 0 && rad.y == 0 && rad.z == 0) continue;
        ++cnt.shadow;
        c = add(c, scl(mul(f_d, rad), lw));
        c = add(c, scl(mul(f_s, rad), lw));
    }
    return c;
}

// Lanes take vertices in queue order.  A vertex without a record (nothing to add but the ambient term) is finished on the
// spot; the others are collected per wave and evaluated 64 at a time, so that the lanes of the expensive part are all busy
// however the lit and the shadowed vertices are mixed.
constexpr int GATHER_QUEUE = 128;
// LEAN: gather mode 3 as a kernel of its own (what it leaves out is most of the code, and the registers that code would claim).
template <bool LEAN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_GATHER_WAVES))) light_gather_kernel(DevScene S, Wavefront W, int gen, int pass, const int *__restrict__ light_of, int k_total,
                                                            int k_pad, int slot_table, int has_records, int defer,
                                                            unsigned long long *__restrict__ stats) {
    __shared__ unsigned pending[4][GATHER_QUEUE];
    const int shard = blockIdx.x % N_SHARDS;
    const unsigned shard_base = shard * W.shard_entries, shadow_base = shard * W.shadow_shard_entries;
    unsigned nv = *counter_of(W, Q_VERTEX, gen, shard);
    if (nv > W.shard_entries) nv = W.shard_entries;
    const unsigned lane = threadIdx.x & 63u;
    volatile unsigned *mine = pending[threadIdx.x >> 6];   // volatile: written and read by different lanes of the wave
    const unsigned wave_global = ((blockIdx.x / N_SHARDS) * 256u + threadIdx.x) >> 6;
    const unsigned n_waves = ((gridDim.x / N_SHARDS) * 256u) >> 6;
    const int k0 = pass * 64, kn = has_records ? min(k_total - k0, 64) : 0;
    const VertexBuf &B = W.vtx[gen & 1];
    Counters cnt;
    memset(&cnt, 0, sizeof cnt);
    auto finish = [&](V c, float4 w, int slot_scale) {
        if (slot_scale < 0) c = scl(c, 0.004f);
        accum_add(W, slot_scale & 0x7fffffff, mul(mk(w.x, w.y, w.z), c));
    };

    auto evaluate = [&](unsigned n_ready) {   // the first n_ready (<= 64) collected vertices
        const bool on = lane < n_ready;
        const unsigned vl = mine[on ? lane : 0u];
        const unsigned v = shard_base + vl;
        if (LEAN) {
            const float4 fk = B.n_key[v], fd2 = B.view_mat[v], ws = B.kd_slot[v], dt = W.shadow[gen & 1].d_time[shadow_base + vl];
            const V c = gather_lean(S, on, mk(fk.x, fk.y, fk.z), mk(fd2.x, fd2.y, fd2.z), mk(dt.x, dt.y, dt.z), fd2.w, __float_as_uint(fk.w), cnt);
            if (on) finish(c, ws, __float_as_int(ws.w));
            return;
        }
        const float4 af = B.amb_first[v], w = B.weight[v];
        const unsigned long long have = (unsigned long long)(kn > 32 ? __float_as_uint(w.w) : 0u) << 32 | __float_as_uint(af.w);
        V c = pass == 0 ? mk(af.x, af.y, af.z) : mk(0, 0, 0);
        c = gather_vertex(S, W, gen, v, vl, shadow_base, light_of, k_total, k0, kn, k_pad, slot_table, defer, on, on ? have : 0ull, c, cnt);
        if (on) finish(c, w, __float_as_int(B.kd_slot[v].w));
    };
    unsigned n_pending = 0;   // uniform over the wave
    for (unsigned vbase = wave_global * 64u; vbase < nv; vbase += n_waves * 64u) {
        const unsigned vl = vbase + lane;
        const bool valid = vl < nv;
        const unsigned v = shard_base + (valid ? vl : vbase);
        // defer 3: as 2, and the vertices have neither an ambient term nor a light-sample record: one the emitter's ray does
        // not reach adds exactly nothing (its sum is throughput x 0), so all that is read of it is the ray's outcome
        constexpr bool lean = LEAN;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 af = lean ? zero4 : B.amb_first[v], w = lean ? zero4 : B.weight[v];
        unsigned long long have = has_records && !lean ? ((unsigned long long)(kn > 32 ? __float_as_uint(w.w) : 0u) << 32 | __float_as_uint(af.w)) : 0ull;
        const bool room = record_index(shadow_base, vbase, (unsigned)k_pad, (unsigned)k_pad - 1u, 63u) < shadow_base + W.shadow_shard_entries;
        if (!room) {   // (light_setup_kernel has reported it; with the rays written by the shading kernel this is where it shows)
            have = 0;
            *W.overflow = 1u;
        }
        bool full = valid && have != 0;
        if (defer != 0 && room) {   // `defer`: the pass holds deferred lights; is one of them visible from the vertex?
            for (int kl = 0, kend; kl < kn; kl = kend) {
                const int li = light_of[k0 + kl];
                const rt_light &L = S.lights[li];
                kend = min(kn, light_of[k_total + li] + L.samples - k0);
                if (L.type != RT_LIGHT_SPHERE) continue;
                if (defer >= 2) {
                    if (valid && W.shadow[gen & 1].vis[shadow_base + vl] != 0.f) full = true;
                    continue;
                }
                const unsigned ri = record_index(shadow_base, vbase, (unsigned)k_pad, record_slot(light_of, slot_table, (unsigned)(k0 + kl), kl), 0u);
                const unsigned first_ray = __float_as_uint(((const float *)&W.shadow[gen & 1].fd_ray[ri])[3]);
                if (valid && W.shadow[gen & 1].vis[first_ray + lane] != 0.f) full = true;
            }
        }
        if (valid && !full && !lean) finish(pass == 0 ? mk(af.x, af.y, af.z) : mk(0, 0, 0), w, __float_as_int(B.kd_slot[v].w));
        const unsigned long long mask = __ballot(full);
        if (mask) {
            const unsigned at = n_pending + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (full) mine[at] = vl;
            n_pending += (unsigned)__popcll(mask);
            if (n_pending >= 64u) {
                evaluate(64u);
                const unsigned keep = lane + 64u < n_pending ? mine[lane + 64u] : 0u;
                mine[lane] = keep;
                n_pending -= 64u;
            }
        }
    }
    if (n_pending) evaluate(n_pending);
    flush_counters(cnt, stats, false);
}

// ---- resolve: per-sample clamp (src/ray_trace.cpp:1158-1167) and the pixel mean (src/main.cpp:464-483) --------------
// `packed`: the frame goes out tile by tile, tile t of the rank's list at floats [t * 3072, (t + 1) * 3072), pixels row-major
// inside the tile (what the ranks of a multi-GPU frame send to the root); otherwise at its place in the full frame.
__global__ void __launch_bounds__(256) resolve_kernel(rt_camera_desc cam, Batch b, Wavefront W, float *__restrict__ out, int packed) {
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= b.n_pixels) return;
    int i, j;
    batch_pixel(b, pl, i, j);
    if (i >= cam.resx || j >= cam.resy) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int s = 0; s < b.n_samples; ++s) {
        const float *src = W.accum + ((size_t)s * b.n_pixels + pl);
        float r = src[0], g = src[(size_t)W.capacity], bl = src[2 * (size_t)W.capacity];
        if (cam.sample_max_val > 0.0f) {
            r = min_std(r, cam.sample_max_val);
            g = min_std(g, cam.sample_max_val);
            bl = min_std(bl, cam.sample_max_val);
        }
        a0 += r;
        a1 += g;
        a2 += bl;
    }
    const float ns = (float)b.n_samples;
    float *dst = packed ? out + ((size_t)(b.tile_first + (pl >> 10)) * (TILE * TILE) + (size_t)((j % TILE) * TILE + (i % TILE))) * 3
                        : out + ((size_t)j * cam.resx + i) * 3;
    dst[0] = rt_div(a0, ns);
    dst[1] = rt_div(a1, ns);
    dst[2] = rt_div(a2, ns);
}

// Primary visibility read-out: the closest-hit records of generation 0, written per pixel.
__global__ void __launch_bounds__(256) primary_out_kernel(DevScene S, rt_camera_desc cam, Batch b, Wavefront W, int32_t *__restrict__ shape_out,
                                                           int32_t *__restrict__ id_out, float *__restrict__ t_out) {
    const int shard = blockIdx.x % N_SHARDS;
    const unsigned shard_base = shard * W.shard_entries;
    unsigned count = *counter_of(W, Q_PATH, 0, shard);
    if (count > W.shard_entries) count = W.shard_entries;
    for (unsigned li = (blockIdx.x / N_SHARDS) * 256u + threadIdx.x; li < count; li += (gridDim.x / N_SHARDS) * 256u) {
        const unsigned idx = shard_base + li;
        const unsigned slot = __float_as_uint(W.q[0].thr_slot[idx].w);
        int i, j;
        batch_pixel(b, (int)(slot % (unsigned)b.n_pixels), i, j);
        const size_t px = (size_t)j * cam.resx + i;
        const float2 found = W.hit[idx];
        const int prim = __float_as_int(found.y);
        int shape = -1, id = -1;
        if (prim <= -2) {
            shape = S.n_triangles + S.n_spheres + (-2 - prim);
        } else if (prim >= 0) {
            const int4 meta = S.prim_meta[2 * (size_t)prim];
            shape = meta.y;
            id = meta.z;
        }
        shape_out[px] = shape;
        id_out[px] = id;
        t_out[px] = prim == -1 ? 1e30f : found.x;
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

// A scene's buffers come from the device's block cache (rt_devmem.h) and go back to it in rt_scene_destroy.
template <class T> int upload(const T *src, size_t n, T **dst) {
    *dst = (T *)device_blocks().alloc(std::max<size_t>(n, 1) * sizeof(T));   // (pointers stay valid for empty arrays)
    if (!*dst) {
        g_error = "out of device memory for the scene";
        return 10;
    }
    if (src && n > 0) {
        if (hipError_t e = hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice); e != hipSuccess) {
            g_error = std::string("hipMemcpy: ") + hipGetErrorString(e);
            device_blocks().free(*dst);
            *dst = nullptr;
            return 10;
        }
    }
    return 0;
}

}  // namespace

struct rt_scene {
    DevScene dev{};
    std::vector<void *> allocations;
    unsigned long long *stats = nullptr;   // STAT_ROWS x STAT_WORDS device counters, followed by a copy taken before the current batch
    float *framebuffer = nullptr;
    size_t framebuffer_floats = 0;
    bool counters = false;
    // The wide tree takes a few launches over the nodes to build and saves device time per ray, so it is built when the
    // scene has been asked for enough rays to repay it; until then the two-box records serve.
    // RT_WIDE_NODES=1: at scene creation, =0: never.
    rt_bvh_node *tree_nodes = nullptr;     // the scene's node array on the device, kept while the wide tree may still be built
    int tree_node_count = 0;
    std::vector<int> wide_top;             // the nodes that head the first wide records, breadth-first (see build_wide_nodes in rt_wide.h)
    unsigned long long paths_seen = 0;
    int device = 0;
    hipStream_t stream = nullptr;    // generate, closest-hit trace, shade, resolve
    hipStream_t stream2 = nullptr;   // direct lighting (light setup, any-hit trace, gather)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> shaded, lit;   // per generation: shade done (stream) / direct lighting done (stream2)
    // wavefront state (allocated on first render, kept)
    Wavefront wf{};
    std::vector<void *> wf_allocations;
    unsigned *host_flags = nullptr;   // pinned: overflow flag read back per batch
    int *light_of = nullptr;          // [k_total] light-sample number -> light index, [n_lights] light -> its first sample number, [k_total] slot table
    int k_total = 0, k_pad = 1;       // light samples per shading point; record slots per vertex with every sample recorded
    int k_pad_deferred = 1;           // record slots per vertex with sphere emitters deferred (see record_slot)
    int slot_table = -1;              // offset of the deferred-mode slot table in light_of
    std::vector<char> pass_has_emitter;   // per pass of 64 light samples: does it hold samples of a sphere emitter
    bool has_dielectric = false;
    bool only_regular_materials = true;   // no mirror, conductor or dielectric: recursive ray tracing (trace_rec) never spawns a ray
    bool plain_shading = false;           // and no texture either: the shading kernel without the code for them (RT_PLAIN_SHADING=0: the full kernel)
    bool single_sphere_emitter = false;   // the scene's lights are exactly one sphere emitter
    // this rank's tiles of a frame that is split over several ranks (see shard_owner)
    int *owned_tiles = nullptr;       // device copy of the list
    size_t owned_room = 0;
    int owned_key[4] = {0, 0, 0, 0};  // resx, resy, rank, world the list was built for
    int n_owned = 0;
    unsigned trace_blocks[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // resident workgroups of trace_kernel<ANY, ...> (plain, counting, wide, wide in any order), a multiple of N_SHARDS
    bool shadow_order_free = false;   // see shadow_answers_order_free
    // per-kernel timing (rt_set_profiling): every launch bracketed by two events on `stream`
    bool profiling = false;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    struct Bracket {
        int kernel;
        hipEvent_t begin, end;
    };
    std::vector<Bracket> brackets;
    double kernel_ms[RT_KERNEL_COUNT] = {};
    unsigned long long kernel_launches[RT_KERNEL_COUNT] = {};
    // A frame that fits one batch is some forty short launches (1080p / 1 spp: 1.1 ms of kernels): when the same frame is
    // asked for again -- same camera, same buffers -- its launch sequence is captured once into a HIP graph and replayed.
    bool frame_graph_off = false;             // a capture failed: no further attempts for this scene
    unsigned long long frame_key_seen = 0;    // key of the last single-batch frame that was launched kernel by kernel
    unsigned long long frame_graph_key = 0;   // key the graph below was captured for (0: none)
    hipGraph_t frame_graph = nullptr;
    hipGraphExec_t frame_graph_exec = nullptr;
};

static void drop_frame_graph(rt_scene *s) {
    if (s->frame_graph_exec) (void)hipGraphExecDestroy(s->frame_graph_exec);
    if (s->frame_graph) (void)hipGraphDestroy(s->frame_graph);
    s->frame_graph_exec = nullptr;
    s->frame_graph = nullptr;
    s->frame_graph_key = 0;
}

// Times one kernel launch when profiling is on: records an event before and (on destruction) after it.
struct LaunchTimer {
    rt_scene *s;
    hipStream_t stream;
    hipEvent_t end = nullptr;
    LaunchTimer(rt_scene *scene, int kernel, hipStream_t on) : s(scene), stream(on) {
        if (!s->profiling) return;
        while (s->event_pool.size() < s->events_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            s->event_pool.push_back(e);
        }
        hipEvent_t begin = s->event_pool[s->events_used++];
        end = s->event_pool[s->events_used++];
        (void)hipEventRecord(begin, stream);
        s->brackets.push_back({kernel, begin, end});
    }
    ~LaunchTimer() {
        if (end) (void)hipEventRecord(end, stream);
    }
};

// After the stream has been synchronised: adds the bracketed launches to the per-kernel totals.
static void harvest_timers(rt_scene *s) {
    for (const auto &b : s->brackets) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, b.begin, b.end) == hipSuccess) {
            s->kernel_ms[b.kernel] += ms;
            ++s->kernel_launches[b.kernel];
            static const bool verbose = getenv("RT_PROFILE_LAUNCHES") != nullptr;   // diagnostic: one line per launch
            if (verbose) fprintf(stderr, "launch kernel %d: %.3f ms\n", b.kernel, ms);
        }
    }
    s->brackets.clear();
    s->events_used = 0;
}

static size_t counter_words(int generations) { return (size_t)3 * generations * N_SHARDS * COUNTER_STRIDE + 1; }   // append counters + overflow flag

// Upper bound of entries per path queue / vertex buffer / accumulator.  An entry costs 348 bytes plus 136 bytes per
// shadow slot (`shadow_mult` slots per entry: rays and light-sample records of one vertex in one lighting pass, both
// generations' sets); the default is what fits 128 GB of the 288 GB -- 264 Mi entries at one slot per vertex, 50 Mi at
// sixteen -- and RT_WAVEFRONT_CAPACITY overrides it.  Large batches matter: every kernel of a batch ends with a tail in
// which the last rays finish on a mostly idle chip, and the fewer batches a frame has, the fewer tails it pays for
// (4K / 64 spp frame of the 1M-triangle scene: 948 ms at 12 Mi entries, 857 at 24 Mi, 803 at 48 Mi, 775 at 96 Mi, 740 at 264 Mi).
static unsigned wavefront_capacity(unsigned shadow_mult, bool records = true) {
    const unsigned long long unit = N_SHARDS * 256;
    if (const char *e = getenv("RT_WAVEFRONT_CAPACITY")) {
        long v = atol(e);
        if (v >= 65536 && v <= (1l << 28)) return (unsigned)((unsigned long long)v / unit * unit);
    }
    unsigned long long budget = 128000000000ull;
    size_t free_bytes = 0, total_bytes = 0;   // share a device that is already in use: at most 3/4 of what is free
    if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) budget = std::min<unsigned long long>(budget, (unsigned long long)free_bytes / 4 * 3);
    (void)hipGetLastError();
    const unsigned long long per_entry = 348ull + (records ? 136ull : 72ull) * shadow_mult;
    return (unsigned)(std::max<unsigned long long>(std::min<unsigned long long>(budget / per_entry, 1ull << 28) / unit, 4) * unit);
}

// Allocates the wavefront state for `cap` entries; on failure everything allocated so far is released.
static int allocate_wavefront(rt_scene *s, unsigned cap, unsigned shadow_mult, int generations, bool records, Wavefront &W) {
    auto alloc = [&](void **p, size_t bytes) -> bool {
        if (hipMalloc(p, bytes) != hipSuccess) return false;
        s->wf_allocations.push_back(*p);
        return true;
    };
    const size_t scap = (size_t)cap * shadow_mult;   // shadow rays and records: `shadow_mult` per vertex at full path capacity
    bool ok = true;
    for (int k = 0; k < 2 && ok; ++k) {
        ok = alloc((void **)&W.q[k].o_time, (size_t)cap * 16) && alloc((void **)&W.q[k].d_key, (size_t)cap * 16) &&
             alloc((void **)&W.q[k].thr_slot, (size_t)cap * 16) && alloc((void **)&W.q[k].depth_flags, (size_t)cap * 4) &&
             alloc((void **)&W.vtx[k].p_time, (size_t)cap * 16) && alloc((void **)&W.vtx[k].n_key, (size_t)cap * 16) &&
             alloc((void **)&W.vtx[k].view_mat, (size_t)cap * 16) && alloc((void **)&W.vtx[k].kd_slot, (size_t)cap * 16) &&
             alloc((void **)&W.vtx[k].ks_n, (size_t)cap * 16) && alloc((void **)&W.vtx[k].weight, (size_t)cap * 16) &&
             alloc((void **)&W.vtx[k].amb_first, (size_t)cap * 16) && alloc((void **)&W.shadow[k].o_dist, scap * 16) &&
             alloc((void **)&W.shadow[k].d_time, scap * 16) && alloc((void **)&W.shadow[k].vis, scap * 4) &&
             (!records || (alloc((void **)&W.shadow[k].fd_ray, scap * 16) && alloc((void **)&W.shadow[k].fs_, scap * 16)));
    }
    ok = ok && alloc((void **)&W.hit, (size_t)cap * 8) &&
         alloc((void **)&W.accum, (size_t)cap * 12) && alloc((void **)&W.counters, counter_words(generations) * sizeof(unsigned));
    if (!ok) {
        for (void *p : s->wf_allocations) (void)hipFree(p);
        s->wf_allocations.clear();
        (void)hipGetLastError();
        return 10;
    }
    W.shadow_shard_entries = (unsigned)(scap / N_SHARDS);
    W.has_records = records ? 1 : 0;
    W.generations = generations;
    W.overflow = W.counters + counter_words(generations) - 1;
    W.shard_entries = cap / N_SHARDS;
    W.capacity = cap;
    return 0;
}

// Idle wavefront states are kept when a scene is destroyed -- up to IDLE_WAVEFRONTS per device, as many as scenes may be alive
// on it at once in a sequence of frames (one scene per file, a few render threads) -- and handed to the next scenes created
// on that device: such a sequence would otherwise allocate and free gigabytes per frame, and the driver answers that with
// occasional stalls of seconds.
struct IdleWavefront {
    Wavefront wf{};
    std::vector<void *> allocations;
};
constexpr size_t IDLE_WAVEFRONTS = 4;
constexpr unsigned long long IDLE_WAVEFRONT_BYTES = 32ull << 30;   // what several parked states may hold together (one alone may be larger)
std::mutex g_idle_mutex;
std::map<int, std::vector<IdleWavefront>> g_idle_wavefronts;

static void adopt_idle_wavefront(rt_scene *s) {   // the largest one parked on the scene's device
    std::lock_guard<std::mutex> lk(g_idle_mutex);
    auto it = g_idle_wavefronts.find(s->device);
    if (it == g_idle_wavefronts.end() || it->second.empty()) return;
    std::vector<IdleWavefront> &idle = it->second;
    size_t best = 0;
    for (size_t k = 1; k < idle.size(); ++k)
        if (idle[k].wf.capacity > idle[best].wf.capacity) best = k;
    s->wf = idle[best].wf;
    s->wf_allocations = std::move(idle[best].allocations);
    idle.erase(idle.begin() + (long)best);
}

// true when the scene's wavefront state was parked (the scene no longer owns it)
static bool park_wavefront(rt_scene *s) {
    if (!s->wf.capacity) return false;
    std::lock_guard<std::mutex> lk(g_idle_mutex);
    std::vector<IdleWavefront> &idle = g_idle_wavefronts[s->device];
    auto bytes_of = [](const Wavefront &W) {   // (allocate_wavefront's sizes)
        return (unsigned long long)W.capacity * 348ull + (unsigned long long)W.shadow_shard_entries * N_SHARDS * (W.has_records ? 136ull : 72ull);
    };
    unsigned long long parked_bytes = 0;
    for (const IdleWavefront &w : idle) parked_bytes += bytes_of(w.wf);
    if (idle.size() >= IDLE_WAVEFRONTS || (!idle.empty() && parked_bytes + bytes_of(s->wf) > IDLE_WAVEFRONT_BYTES)) {
        // no room: this one takes the place of the smallest, if it is larger
        size_t least = 0;
        for (size_t k = 1; k < idle.size(); ++k)
            if (idle[k].wf.capacity < idle[least].wf.capacity) least = k;
        if (idle[least].wf.capacity >= s->wf.capacity) return false;
        for (void *p : idle[least].allocations) (void)hipFree(p);
        idle.erase(idle.begin() + (long)least);
    }
    IdleWavefront parked;
    parked.wf = s->wf;
    parked.allocations = std::move(s->wf_allocations);
    idle.push_back(std::move(parked));
    s->wf = Wavefront{};
    s->wf_allocations.clear();
    return true;
}

// The two streams and the events of a destroyed scene wait here for the next scene on the same device (up to IDLE_STREAM_SETS a
// device): making and destroying them was a quarter of a millisecond per scene of a frame sequence.  Everything on the streams
// has finished when a set is parked.
struct StreamSet {
    hipStream_t stream = nullptr, stream2 = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> shaded, lit;
    unsigned *host_flags = nullptr;   // the pinned word a frame's overflow flag is copied to (rt_scene::host_flags), made on first use
};
constexpr size_t IDLE_STREAM_SETS = 8;
std::mutex g_stream_sets_mutex;
std::map<int, std::vector<StreamSet>> g_idle_stream_sets;

static void destroy_stream_set(StreamSet &t) {
    for (hipEvent_t e : t.shaded) (void)hipEventDestroy(e);
    for (hipEvent_t e : t.lit) (void)hipEventDestroy(e);
    if (t.stream2) (void)hipStreamDestroy(t.stream2);
    if (t.ev0) (void)hipEventDestroy(t.ev0);
    if (t.ev1) (void)hipEventDestroy(t.ev1);
    if (t.stream) (void)hipStreamDestroy(t.stream);
    if (t.host_flags) (void)hipHostFree(t.host_flags);
    t = StreamSet();
}

// A parked set of the current scene's device, or a new one; false when the runtime cannot make one.
static bool take_stream_set(rt_scene *s) {
    StreamSet t;
    {
        std::lock_guard<std::mutex> lk(g_stream_sets_mutex);
        std::vector<StreamSet> &idle = g_idle_stream_sets[s->device];
        if (!idle.empty()) {
            t = std::move(idle.back());
            idle.pop_back();
        }
    }
    if (!t.stream && (hipStreamCreate(&t.stream) != hipSuccess || hipStreamCreate(&t.stream2) != hipSuccess || hipEventCreate(&t.ev0) != hipSuccess ||
                      hipEventCreate(&t.ev1) != hipSuccess)) {
        destroy_stream_set(t);
        return false;
    }
    s->stream = t.stream, s->stream2 = t.stream2, s->ev0 = t.ev0, s->ev1 = t.ev1;
    s->shaded = std::move(t.shaded), s->lit = std::move(t.lit);
    s->host_flags = t.host_flags;
    return true;
}

static void park_stream_set(rt_scene *s) {   // (the scene's streams are idle)
    StreamSet t;
    t.stream = s->stream, t.stream2 = s->stream2, t.ev0 = s->ev0, t.ev1 = s->ev1;
    t.shaded = std::move(s->shaded), t.lit = std::move(s->lit);
    t.host_flags = s->host_flags;
    s->host_flags = nullptr;
    s->stream = s->stream2 = nullptr;
    s->ev0 = s->ev1 = nullptr;
    s->shaded.clear(), s->lit.clear();
    if (t.stream && t.stream2 && t.ev0 && t.ev1 && t.shaded.size() <= 64) {
        std::lock_guard<std::mutex> lk(g_stream_sets_mutex);
        std::vector<StreamSet> &idle = g_idle_stream_sets[s->device];
        if (idle.size() < IDLE_STREAM_SETS) {
            idle.push_back(std::move(t));
            return;
        }
    }
    destroy_stream_set(t);
}

// Makes the wavefront state hold at least min(`wanted`, the capacity bound) entries and `shadow_floor` shadow slots
// (what one batch of the smallest size needs); grows, never shrinks.  A new state gets `shadow_mult` shadow slots per
// entry.  When the device cannot give that much, half of it is tried, and so on down to 1 Mi entries.
static int ensure_wavefront(rt_scene *s, unsigned long long wanted, unsigned shadow_mult, unsigned long long shadow_floor, int generations, bool records = true) {
    // generate deals whole 256-lane workgroups to the shards, so a shard must hold ceil(workgroups / N_SHARDS) of them
    const unsigned long long unit = N_SHARDS * 256;
    wanted = std::max<unsigned long long>(65536, (wanted + unit - 1) / unit * unit);
    unsigned cap = (unsigned)std::min<unsigned long long>(wanted, wavefront_capacity(shadow_mult, records));
    auto enough = [&](const Wavefront &W) {
        return W.capacity >= cap && W.generations >= generations && (W.has_records || !records) &&
               (unsigned long long)W.shadow_shard_entries * N_SHARDS >= std::min<unsigned long long>(shadow_floor, (unsigned long long)cap * shadow_mult);
    };
    if (enough(s->wf)) return 0;
    if (!s->host_flags) HIP_OK(hipHostMalloc((void **)&s->host_flags, sizeof(unsigned)));
    if (!s->wf.capacity) {
        adopt_idle_wavefront(s);
        if (enough(s->wf)) return 0;
    }
    HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipStreamSynchronize(s->stream2));
    for (void *p : s->wf_allocations) (void)hipFree(p);
    s->wf_allocations.clear();
    s->wf = Wavefront{};
    for (;;) {
        Wavefront W{};
        if (allocate_wavefront(s, cap, shadow_mult, std::max(generations, 8), records, W) == 0) {
            s->wf = W;
            return 0;
        }
        if (cap <= (1u << 20)) break;
        cap = (unsigned)((cap / 2 + unit - 1) / unit * unit);
    }
    g_error = "cannot allocate the wavefront state (lower RT_WAVEFRONT_CAPACITY)";
    return 10;
}

// Launches the persistent trace kernel: as many workgroups as stay resident at once, never more than `grid_limit`
// (the grid a one-ray-per-lane launch would use; small batches need fewer workgroups than the chip holds).
static void launch_trace(rt_scene *s, hipStream_t stream, bool any, unsigned grid_limit, int gen, int whitted) {
    const bool count = s->counters;
    // the wide tree, once built, serves every render without counters
    const bool wide = !count && s->dev.wnodes != nullptr;
    typedef void (*TraceFn)(DevScene, Wavefront, int, int, TraceTuning, unsigned long long *);
    const bool order_free = wide && any && s->shadow_order_free;
    const TraceFn fn = order_free ? trace_kernel<true, false, true, true>
                       : wide  ? (any ? trace_kernel<true, false, true> : trace_kernel<false, false, true>)
                       : count ? (any ? trace_kernel<true, true, false> : trace_kernel<false, true, false>)
                               : (any ? trace_kernel<true, false, false> : trace_kernel<false, false, false>);
    unsigned &blocks = s->trace_blocks[any][order_free ? 3 : (wide ? 2 : (count ? 1 : 0))];
    if (!blocks) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device) != hipSuccess || cus < 1) cus = 256;
        (void)hipGetLastError();
        blocks = std::max(1u, (unsigned)per_cu * (unsigned)cus / N_SHARDS) * N_SHARDS;
    }
    const dim3 grid(std::min(blocks, std::max(grid_limit, (unsigned)N_SHARDS)));
    LaunchTimer timer(s, any ? RT_KERNEL_TRACE_ANY : RT_KERNEL_TRACE_CLOSEST, stream);
    static const TraceTuning tune = [] {
        TraceTuning t{16, 16, 8};   // (closest hit of a 4K / 64 spp frame with pop_min 1 / 4 / 8 / 16: 253.2 / 247.4 / 245.4 / 245.9 ms)
        if (const char *e = getenv("RT_TRACE_POP_MIN")) t.pop_min = std::min(64, std::max(1, atoi(e)));
        if (const char *e = getenv("RT_TRACE_REFILL_MIN")) t.refill_min = std::min(64, std::max(1, atoi(e)));
        if (const char *e = getenv("RT_TRACE_LEAF_MIN")) t.leaf_min = std::min(64, std::max(1, atoi(e)));
        return t;
    }();
    hipLaunchKernelGGL(fn, grid, dim3(256), 0, stream, s->dev, s->wf, gen, whitted, tune, s->stats);
}

// Does the answer of the scene's shadow rays depend on the order in which a ray's walk looks at the leaves?  A shadow ray
// ends with the first hit before the light sample (t < dist), and counts as shadow when that hit lies more than the
// shadow-ray epsilon before it (t < dist - epsilon, src/ray_trace.cpp:232-241).  If every hit a ray can find is of the
// second kind, "some hit exists" decides, whatever is found first; if a surface can lie within epsilon of the sample -- an
// emitter's own triangle, on which its samples sit -- the answer depends on whether that surface or a real occluder is
// found first, and only the reference's order gives the reference's answer.  So: true when no leaf of the tree that
// comes within reach of a light's sample positions (a point for point, spot and sphere lights -- every sample of a sphere
// light aims at its centre, src/lights.cpp:418-422 --, a parallelogram for area lights; directional and environment lights
// end at no surface) holds a triangle, and the spheres in such leaves keep their surface away from the point.  Planes are
// tested before the tree by every walk.  Conservative: a scene that fails keeps the ordered walk.
static bool shadow_answers_order_free(const rt_scene_desc *d) {
    // a scene that comes without a tree (one is built on the device) is judged by its primitives' own boxes: every leaf of
    // whatever tree is built lies inside the boxes of the primitives it holds
    const bool by_primitives = (!d->bvh_nodes || d->n_bvh_nodes < 1) && d->bvh_prim_bounds && d->n_bvh_prims > 0;
    if (!by_primitives && (!d->bvh_nodes || d->n_bvh_nodes < 1)) return false;
    if (!d->bvh_prims) return false;
    const float eps = std::fabs(d->shadow_ray_epsilon);
    for (int li = 0; li < d->n_lights; ++li) {
        const rt_light &L = d->lights[li];
        float lo[3], hi[3];
        bool point = true;
        if (L.type == RT_LIGHT_DIRECTIONAL || L.type == RT_LIGHT_ENV) continue;
        if (L.type == RT_LIGHT_TRIANGLE) return false;
        if (L.type == RT_LIGHT_POINT || L.type == RT_LIGHT_SPOT) {
            for (int a = 0; a < 3; ++a) lo[a] = hi[a] = L.position[a];
        } else if (L.type == RT_LIGHT_SPHERE) {
            if (L.sphere < 0 || L.sphere >= d->n_spheres) return false;
            const float *m = d->spheres[L.sphere].model;   // the sample position: model * (0, 0, 0, 1)
            for (int a = 0; a < 3; ++a) lo[a] = hi[a] = m[4 * a + 3];
        } else if (L.type == RT_LIGHT_AREA) {
            point = false;
            for (int a = 0; a < 3; ++a) {
                const float r = 0.5f * std::fabs(L.size) * (std::fabs(L.u[a]) + std::fabs(L.v[a]));
                lo[a] = L.position[a] - r, hi[a] = L.position[a] + r;
            }
        } else {
            return false;
        }
        float reach = 0.f;
        for (int a = 0; a < 3; ++a) reach = std::max(reach, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
        reach = 4.0f * eps + 1e-4f * reach + 1e-3f;   // epsilon, rounding of the distances, padding of the boxes
        if (!std::isfinite(reach)) return false;
        auto apart = [&](const float *bmin, const float *bmax) {
            bool yes = false;
            for (int a = 0; a < 3; ++a) yes |= bmin[a] > hi[a] + reach || bmax[a] < lo[a] - reach;   // (a box that is not a number is not apart)
            return yes;
        };
        auto keeps_away = [&](const rt_bvh_prim &pr) {   // a primitive whose box comes within reach
            if (pr.shape < d->n_triangles || !point) return false;
            // a sphere: distance of the point from its surface, in its own space, brought to the world by the least
            // stretch of its transform (1 / Frobenius norm of the inverse bounds that from below)
            const rt_sphere &sp = d->spheres[pr.shape - d->n_triangles];
            const float *im = sp.inv_model;
            float q[3], norm2 = 0.f, off2 = 0.f;
            for (int r = 0; r < 3; ++r) {
                q[r] = im[4 * r] * lo[0] + im[4 * r + 1] * lo[1] + im[4 * r + 2] * lo[2] + im[4 * r + 3];
                for (int c = 0; c < 3; ++c) norm2 += im[4 * r + c] * im[4 * r + c];
                off2 += (q[r] - sp.center[r]) * (q[r] - sp.center[r]);
            }
            const float from_surface = std::fabs(std::sqrt(off2) - sp.radius) / std::sqrt(norm2);
            return from_surface > reach;
        };
        if (by_primitives) {
            // (first a count without branches, which the compiler vectorises: lights mostly hang free of everything)
            const float lo_x = lo[0] - reach, lo_y = lo[1] - reach, lo_z = lo[2] - reach, hi_x = hi[0] + reach, hi_y = hi[1] + reach, hi_z = hi[2] + reach;
            const float *b = d->bvh_prim_bounds;
            int near = 0;
            for (int i = 0; i < d->n_bvh_prims; ++i, b += 6)
                near += !((b[0] > hi_x) | (b[1] > hi_y) | (b[2] > hi_z) | (b[3] < lo_x) | (b[4] < lo_y) | (b[5] < lo_z)) ? 1 : 0;
            if (near == 0) continue;
            for (int i = 0; i < d->n_bvh_prims; ++i) {
                const float *bi = d->bvh_prim_bounds + 6 * (size_t)i;
                if (!apart(bi, bi + 3) && !keeps_away(d->bvh_prims[i])) return false;
            }
            continue;
        }
        std::vector<int> todo{0};
        while (!todo.empty()) {
            const int i = todo.back();
            todo.pop_back();
            const rt_bvh_node &n = d->bvh_nodes[i];
            if (apart(n.bmin, n.bmax)) continue;
            if (n.n_prims == 0) {
                todo.push_back(i + 1);
                todo.push_back(n.offset);
                continue;
            }
            for (int k = 0; k < n.n_prims; ++k)
                if (!keeps_away(d->bvh_prims[n.offset + k])) return false;
        }
    }
    return true;
}

// RT_CREATE_TIMING: stage times of rt_scene_create and rt_scene_destroy on stderr (diagnostic)
struct CreateClock {
    const bool on = getenv("RT_CREATE_TIMING") != nullptr;
    std::chrono::high_resolution_clock::time_point last = std::chrono::high_resolution_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::high_resolution_clock::now();
        fprintf(stderr, "[create] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

// The nodes that head the first wide records: the root and the heads a breadth-first walk meets next, as many as the trace
// kernels keep in LDS (the order build_wide_nodes of rt_wide.h gives the top of the tree).  The root must be interior.
static std::vector<int> wide_top_of(const rt_bvh_node *nodes) {
    std::vector<int> top, walk{0};
    const int most = std::max(TOP_NODES, RT_ANY_TOP_NODES);
    for (size_t at = 0; at < walk.size() && (int)top.size() < most; ++at) {
        const int i = walk[at];
        top.push_back(i);
        int slot[4], ax[3];
        wide_slots(nodes, i, slot, ax);
        for (int k = 0; k < 4; ++k)
            if (slot[k] >= 0 && nodes[slot[k]].n_prims == 0) walk.push_back(slot[k]);
    }
    return top;
}

// Builds the wide tree on the device from the kept copy of the scene's node array (which is released either way).  A tree
// that does not qualify leaves the scene on the two-box records; only a failed allocation or copy is an error.
static int build_wide_tree(rt_scene *s) {
    rt_bvh_node *nodes = s->tree_nodes;
    s->tree_nodes = nullptr;
    if (!nodes) return 0;
    HIP_OK(hipSetDevice(s->device));
    HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipStreamSynchronize(s->stream2));
    DeviceBvh built;
    const int rc = wide_records_of_host_tree(nodes, s->tree_node_count, s->wide_top.data(), (int)s->wide_top.size(), LDS_STACK + SPILL_STACK, built, g_error);
    device_blocks().free(nodes);
    s->wide_top.clear();
    if (rc) return rc;
    if (!built.wide_nodes) return 0;
    s->allocations.push_back(built.wide_nodes);
    s->dev.wroot_ref = 0;
    s->dev.n_wtop = built.n_wide_top;
    s->dev.wnodes = built.wide_nodes;
    return 0;
}

// Called by every render with the number of camera paths it is about to trace: the wide tree is built once the scene
// has been asked for 32 paths per BVH node, about where the build is repaid (the walk of the 1M-triangle scene saves 0.28 ms
// per million paths, that of the 70 K-triangle scene at 1 spp 0.06 ms per million; the build takes a few launches over the
// nodes).  That is at once for the 1M-triangle scene's frames, and with the third 1080p frame of the 70 K one.
static int account_paths(rt_scene *s, unsigned long long paths) {
    if (!s->tree_nodes) return 0;
    s->paths_seen += paths;
    if (s->paths_seen < 32ull * (unsigned long long)s->tree_node_count) return 0;
    return build_wide_tree(s);
}

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
    CreateClock clock;
    // operand checks: every index the kernels follow must be in range
    for (int i = 0; i < d->n_bvh_prims; ++i) {
        const rt_bvh_prim &p = d->bvh_prims[i];
        bool ok = p.shape >= 0 && p.shape < d->n_triangles + d->n_spheres &&
                  (p.shape >= d->n_triangles ? p.shape - d->n_triangles <= PRIM_INDEX_MASK
                                             : (p.instance >= 0 && p.instance < d->n_instances && p.instance <= PRIM_INDEX_MASK));
        if (!ok) {
            g_error = "rt_scene_create: BVH primitive out of range";
            return 3;
        }
    }
    for (int i = 0; i < d->n_bvh_nodes; ++i) {
        const rt_bvh_node &n = d->bvh_nodes[i];
        // (a leaf may hold any number of entries; the reference's own count is 16 bits wide, src/bvh.h:20)
        bool ok = n.n_prims > 0    ? (n.offset >= 0 && (long long)n.offset + n.n_prims <= d->n_bvh_prims)
                  : n.n_prims == 0 ? (n.offset > i && n.offset < d->n_bvh_nodes && i + 1 < d->n_bvh_nodes && n.axis >= 0 && n.axis < 3)
                                   : false;
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

    clock.mark("operand checks");
    const bool build_tree = d->n_bvh_nodes == 0 && d->n_bvh_prims > 0;   // the scene comes without a tree: it is built on the device
    if (build_tree && !d->bvh_prim_bounds) {
        g_error = "rt_scene_create: a scene without BVH nodes needs bvh_prim_bounds";
        return 3;
    }
    rt_scene *s = new rt_scene();
    if (hipError_t e = hipGetDevice(&s->device); e != hipSuccess) {
        g_error = std::string("hipGetDevice: ") + hipGetErrorString(e);
        delete s;
        return 10;
    }
    DevScene &D = s->dev;
    auto keep = [&](void *p) { s->allocations.push_back(p); };

    // BVH -> interior-node records holding both child boxes (see DevScene); leaves become references.  Records and leaf
    // entries are written on the device (records_of_host_tree).  Order of the records: the top of the tree breadth-first (the
    // first TOP_NODES interior nodes a breadth-first walk meets, listed here; the trace kernel keeps them in LDS), then the
    // other interior nodes in the order of the reference's array, which is depth-first (src/bvh.cpp:316-344: a node's left
    // child next to it, a small subtree in a few cache lines).
    std::vector<int> top_interior;
    int n_interior = 0, root_ref = 0;
    {
        for (int i = 0; i < d->n_bvh_nodes; ++i) n_interior += d->bvh_nodes[i].n_prims == 0 ? 1 : 0;
        std::vector<int> walk;
        if (d->n_bvh_nodes > 0 && d->bvh_nodes[0].n_prims == 0) walk.push_back(0);
        for (size_t at = 0; at < walk.size() && (int)top_interior.size() < TOP_NODES; ++at) {
            const int i = walk[at];
            top_interior.push_back(i);
            const int kids[2] = {i + 1, d->bvh_nodes[i].offset};
            for (int kid : kids)
                if (d->bvh_nodes[kid].n_prims == 0) walk.push_back(kid);
        }
        s->dev.n_top = (int)top_interior.size();
        if (d->n_bvh_nodes > 0) root_ref = d->bvh_nodes[0].n_prims == 0 ? 0 : (int)(0x80000000u | (unsigned)d->bvh_nodes[0].offset);
    }
    clock.mark("places of the node records");
    // the inverse of the identity as this scene's inverse routine makes it (see plain_placement): the first such placement's
    float id_scale = 1.0f;
    for (int i = 0; i < d->n_instances; ++i)
        if (d->instances[i].identity) {
            id_scale = d->instances[i].inv_model[0];
            break;
        }
    auto as_float = [](int v) {
        float f;
        memcpy(&f, &v, 4);
        return f;
    };
    std::vector<float4> inst((size_t)d->n_instances * 4);
    for (int i = 0; i < d->n_instances; ++i) {
        const rt_instance &in = d->instances[i];
        int flags = (scaled_identity(in, id_scale) ? 1 : 0) | ((in.motionblur[0] != 0.f || in.motionblur[1] != 0.f || in.motionblur[2] != 0.f) ? 2 : 0);
        for (int r = 0; r < 3; ++r) inst[4 * i + r] = make_float4(in.inv_model[4 * r], in.inv_model[4 * r + 1], in.inv_model[4 * r + 2], in.inv_model[4 * r + 3]);
        inst[4 * i + 3] = make_float4(in.motionblur[0], in.motionblur[1], in.motionblur[2], as_float(flags));
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
    clock.mark("placements");
    UP(inst, inst.data(), inst.size());
    UP(instances, d->instances, d->n_instances);
    UP(triangles, d->triangles, d->n_triangles);
    clock.mark("placements and triangles uploaded");
    DeviceBvh built;
    if (build_tree) {
        // the walk's stack holds LDS_STACK + SPILL_STACK far children, one per level; the four-box records are made with the tree
        // (RT_WIDE_NODES=0: not at all)
        const char *wide_mode = getenv("RT_WIDE_NODES");
        const int wide_stack = wide_mode && atoi(wide_mode) == 0 ? 0 : LDS_STACK + SPILL_STACK;
        if (int rc = build_bvh_on_device(d->bvh_prims, d->bvh_prim_bounds, d->n_bvh_prims, D.triangles, d->n_triangles, D.instances, id_scale, LDS_STACK + SPILL_STACK - 1, wide_stack,
                                         built, g_error)) {
            rt_scene_destroy(s);
            return rc;
        }
        root_ref = built.root_ref;
        s->dev.n_top = built.n_top;   // the builder puts the top levels first, as the host's layout does
    } else {
        const char *wide_env = getenv("RT_WIDE_NODES");
        const bool keep_nodes = !(wide_env && atoi(wide_env) == 0) && d->n_bvh_nodes >= 3 && d->bvh_nodes[0].n_prims == 0;
        if (int rc = records_of_host_tree(d->bvh_nodes, d->n_bvh_nodes, top_interior.data(), (int)top_interior.size(), n_interior, d->bvh_prims, d->n_bvh_prims, D.triangles, d->n_triangles,
                                          D.instances, id_scale, keep_nodes, built, g_error)) {
            rt_scene_destroy(s);
            return rc;
        }
        if (built.tree_nodes) {   // the heads of the first wide records, breadth-first, while the nodes are at hand on the host
            s->tree_nodes = built.tree_nodes;
            s->tree_node_count = d->n_bvh_nodes;
            s->wide_top = wide_top_of(d->bvh_nodes);
        }
    }
    D.nodes = built.nodes, D.prims = built.prims, D.prim_meta = built.prim_meta;
    for (void *p : {(void *)built.nodes, (void *)built.prims, (void *)built.prim_meta, (void *)built.wide_nodes})
        if (p) keep(p);
    if (built.wide_nodes) D.wnodes = built.wide_nodes, D.wroot_ref = 0, D.n_wtop = built.n_wide_top;
    if (!built.nodes || !built.prims || !built.prim_meta) {   // (pointers stay valid for an empty array, as upload() keeps them)
        float4 *dummy = nullptr;
        if (upload<float4>(nullptr, 0, &dummy)) {
            rt_scene_destroy(s);
            return 10;
        }
        keep(dummy);
        if (!D.nodes) D.nodes = dummy;
        if (!D.prims) D.prims = dummy;
        if (!D.prim_meta) D.prim_meta = (const int4 *)dummy;
    }
    clock.mark("tree");
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
    clock.mark("materials, textures, lights uploaded");
    D.n_triangles = d->n_triangles;
    D.n_spheres = d->n_spheres;
    D.n_planes = d->n_planes;
    D.n_lights = d->n_lights;
    D.n_nodes = build_tree ? 2 * d->n_bvh_prims - 1 : d->n_bvh_nodes;
    D.root_ref = root_ref;
    if (build_tree) {
        memcpy(D.root_min, built.root_min, sizeof D.root_min);
        memcpy(D.root_max, built.root_max, sizeof D.root_max);
    }
    {
        const char *e = getenv("RT_WIDE_NODES");
        const int mode = e ? atoi(e) : -1;
        if (mode > 0 && build_wide_tree(s) != 0) {
            rt_scene_destroy(s);
            return 10;
        }
    }
    if (d->n_bvh_nodes > 0) {
        memcpy(D.root_min, d->bvh_nodes[0].bmin, sizeof D.root_min);
        memcpy(D.root_max, d->bvh_nodes[0].bmax, sizeof D.root_max);
    }
    D.id_scale = id_scale;
    D.ieps = d->intersection_epsilon;
    D.seps = d->shadow_ray_epsilon;
    D.max_depth = d->max_depth;
    memcpy(D.ambient, d->ambient_light, sizeof D.ambient);
    memcpy(D.bg, d->background_color, sizeof D.bg);
    D.bg_tex = d->background_texture;
    D.is_probe = d->is_probe;
    if (!(s->stats = (unsigned long long *)device_blocks().alloc(2 * STAT_ROWS * STAT_WORDS * sizeof(unsigned long long))) || !take_stream_set(s)) {
        g_error = "rt_scene_create: cannot allocate stream/events";
        rt_scene_destroy(s);
        return 10;
    }
    {   // light-sample numbering of calculate_color's double loop (src/ray_trace.cpp:182-187)
        std::vector<int> light_of;
        for (int i = 0; i < d->n_lights; ++i)
            for (int k = 0; k < d->lights[i].samples; ++k) light_of.push_back(i);
        s->k_total = (int)light_of.size();
        {   // behind the table: the first sample number of every light
            int first = 0;
            for (int i = 0; i < d->n_lights; ++i) {
                light_of.push_back(first);
                first += d->lights[i].samples;
            }
        }
        s->pass_has_emitter.assign((size_t)(s->k_total + 63) / 64, 0);
        for (int k = 0; k < s->k_total; ++k)
            if (d->lights[light_of[(size_t)k]].type == RT_LIGHT_SPHERE) s->pass_has_emitter[(size_t)k / 64] = 1;
        s->k_pad = 1;
        while (s->k_pad < s->k_total && s->k_pad < 64) s->k_pad <<= 1;
        {   // record slots with deferred emitters: per pass, one per other sample and one per emitter
            s->slot_table = (int)light_of.size();
            int most = 1;
            for (int k0 = 0; k0 < s->k_total; k0 += 64) {
                int next = 0, emitter = -1, emitter_slot = 0;
                for (int k = k0; k < std::min(s->k_total, k0 + 64); ++k) {
                    const int li = light_of[(size_t)k];
                    if (d->lights[li].type == RT_LIGHT_SPHERE) {
                        if (li != emitter) {
                            emitter = li;
                            emitter_slot = next++;
                        }
                        light_of.push_back(emitter_slot);
                    } else {
                        light_of.push_back(next++);
                    }
                }
                most = std::max(most, next);
            }
            s->k_pad_deferred = 1;
            while (s->k_pad_deferred < most) s->k_pad_deferred <<= 1;
        }
        if (upload<int>(light_of.data(), light_of.size(), &s->light_of)) {
            rt_scene_destroy(s);
            return 10;
        }
        keep(s->light_of);
        for (int i = 0; i < d->n_materials; ++i) {
            s->has_dielectric |= d->materials[i].type == RT_MAT_DIELECTRIC;
            s->only_regular_materials &= d->materials[i].type == RT_MAT_REGULAR;
        }
        s->single_sphere_emitter = d->n_lights == 1 && d->lights[0].type == RT_LIGHT_SPHERE && d->lights[0].samples <= 64;
        s->plain_shading = s->only_regular_materials && d->n_textures == 0 && !(getenv("RT_PLAIN_SHADING") && atoi(getenv("RT_PLAIN_SHADING")) == 0);
        s->shadow_order_free = !(getenv("RT_ANY_ORDERED") && atoi(getenv("RT_ANY_ORDERED")) != 0) && shadow_answers_order_free(d);
    }
    clock.mark("streams, light tables, decisions about the scene");
    *out = s;
    return 0;
}

void rt_scene_destroy(rt_scene *s) {
    if (!s) return;
    CreateClock clock;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->stream2) (void)hipStreamSynchronize(s->stream2);
    (void)hipSetDevice(s->device);
    drop_frame_graph(s);
    park_wavefront(s);
    clock.mark("destroy: streams, graph, wavefront parked");
    for (void *p : s->allocations) device_blocks().free(p);
    for (void *p : s->wf_allocations) (void)hipFree(p);
    device_blocks().free(s->tree_nodes);
    device_blocks().free(s->stats);
    device_blocks().free(s->framebuffer);
    device_blocks().free(s->owned_tiles);
    for (hipEvent_t e : s->event_pool) (void)hipEventDestroy(e);
    park_stream_set(s);
    clock.mark("destroy: buffers, events, streams released");
    delete s;
}

int rt_set_profiling(rt_scene *s, int32_t enabled) {
    if (!s) return 1;
    s->profiling = enabled != 0;
    for (int k = 0; k < RT_KERNEL_COUNT; ++k) {
        s->kernel_ms[k] = 0.0;
        s->kernel_launches[k] = 0;
    }
    return 0;
}

int rt_kernel_times(const rt_scene *s, rt_kernel_time *out) {
    if (!s || !out) return 1;
    for (int k = 0; k < RT_KERNEL_COUNT; ++k) {
        out[k].launches = s->kernel_launches[k];
        out[k].total_ms = s->kernel_ms[k];
    }
    return 0;
}

int rt_set_counters(rt_scene *s, int32_t enabled) {
    if (!s) return 1;
    s->counters = enabled != 0;
    return 0;
}

// Which rank renders tile (tx, ty) of a frame split over `world` ranks: (tx + skew * ty) mod world, with a skew that is
// coprime to world (the one nearest to 0.38 * world).  Along a row of tiles, and along a column, the ranks follow each
// other cyclically, so every rank's share is spread over the whole image and the shares are even whatever the frame's
// width in tiles; numbering the tiles row by row instead would hand a rank whole columns of tiles whenever the number of
// ranks divides the width.  (The reference hands tiles to its threads dynamically, src/main.cpp:486-512; a static rule
// lets every rank know the layout without a message.)
static inline int shard_skew(int world) {
    auto gcd = [](int a, int b) {
        while (b) {
            const int t = a % b;
            a = b;
            b = t;
        }
        return a;
    };
    int s = std::max(1, (int)(0.381966f * (float)world + 0.5f));
    while (gcd(s, world) != 1) ++s;
    return s;
}
static inline int shard_owner(int tx, int ty, int world) { return (int)(((long long)tx + (long long)shard_skew(world) * ty) % world); }

// Builds (and keeps on the device) the list of tiles `rank` of `world` renders, in row-major order.  With one rank the
// list is implicit.
static int ensure_owned_tiles(rt_scene *s, int resx, int resy, int rank, int world) {
    const int tiles_x = (resx + TILE - 1) / TILE, tiles_y = (resy + TILE - 1) / TILE;
    if (world == 1) {
        s->n_owned = tiles_x * tiles_y;
        return 0;
    }
    const int key[4] = {resx, resy, rank, world};
    if (s->owned_tiles && memcmp(key, s->owned_key, sizeof key) == 0) return 0;
    std::vector<int> list;
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx)
            if (shard_owner(tx, ty, world) == rank) list.push_back(ty * tiles_x + tx);
    if (list.size() + 1 > s->owned_room) {
        HIP_OK(hipStreamSynchronize(s->stream));
        device_blocks().free(s->owned_tiles);
        s->owned_tiles = nullptr;
        s->owned_room = 0;
        if (!(s->owned_tiles = (int *)device_blocks().alloc((list.size() + 1) * sizeof(int)))) {
            g_error = "out of device memory for the tile list";
            return 10;
        }
        s->owned_room = list.size() + 1;
    }
    if (!list.empty()) HIP_OK(hipMemcpy(s->owned_tiles, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice));
    memcpy(s->owned_key, key, sizeof key);
    s->n_owned = (int)list.size();
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
    if (!(s->dev.max_depth <= (float)MAX_DEPTH)) {   // (the reference takes any value, src/parser.cpp:1253-1258; its recursion would not)
        g_error = "MaxRecursionDepth exceeds the supported " + std::to_string(MAX_DEPTH) + " path generations";
        return 5;
    }
    if (cam->splitting_factor > 4096) {
        g_error = "SplittingFactor too large";
        return 5;
    }
    return 0;
}

static void read_stats(rt_scene *s, rt_stats *st, float kernel_ms, float total_ms) {
    if (!st) return;
    unsigned long long rows[STAT_ROWS * STAT_WORDS], v[STAT_USED] = {};
    if (hipMemcpy(rows, s->stats, sizeof rows, hipMemcpyDeviceToHost) != hipSuccess) memset(rows, 0, sizeof rows);
    for (int r = 0; r < STAT_ROWS; ++r)
        for (int k = 0; k < STAT_USED; ++k) v[k] += rows[r * STAT_WORDS + k];
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
    st->shadow_rays_traced = v[9];
    st->any_hit_nodes_visited = v[10];
    st->any_hit_triangle_tests = v[11];
    st->any_hit_sphere_tests = v[12];
    st->any_hit_plane_tests = v[13];
    st->kernel_ms = kernel_ms;
    st->total_ms = total_ms;
}

}  // extern "C"

// Renders the tiles of `rank` of `world` into device memory: at their places in the full frame `dev_out` (resx * resy * 3
// floats; other pixels untouched), or with `packed` tile after tile in the order of the rank's list (3072 floats per tile).
// Returns with the stream drained; kernel_ms = device time between the first and the last kernel.
// The most workgroups generate_kernel deals to one shard for a batch of `per_sample` workgroups per sample and `samples`
// samples: every sample gives each shard per_sample / 64 of them and one more to the per_sample % 64 shards that follow
// the sample's turn (see generate_kernel).
static unsigned long long max_shard_groups(unsigned per_sample, unsigned samples) {
    const unsigned full = per_sample / N_SHARDS, extra = per_sample % N_SHARDS;
    if (!extra || !samples) return (unsigned long long)full * samples;
    if (RT_SHARD_TURN % 2 == 0) return (unsigned long long)(full + 1) * samples;   // (test builds without the turn)
    // whole turns of 64 samples visit every start once: each shard gets `extra` of those
    const unsigned long long rounds = samples / N_SHARDS;
    unsigned count[N_SHARDS] = {0};
    for (unsigned k = 0; k < samples % N_SHARDS; ++k)
        for (unsigned e = 0; e < extra; ++e) ++count[(k * RT_SHARD_TURN + e) % N_SHARDS];
    unsigned most = 0;
    for (unsigned v : count) most = std::max(most, v);
    return (unsigned long long)full * samples + rounds * extra + most;
}

static int render_tiles(rt_scene *s, const rt_camera_desc *cam, int rank, int world, float *dev_out, bool packed, float &kernel_ms) {
    int grid_n, n_samples;
    if (int rc = check_camera(s, cam, grid_n, n_samples)) return rc;
    HIP_OK(hipSetDevice(s->device));
    if (int rc = account_paths(s, (unsigned long long)cam->resx * cam->resy * n_samples / world)) return rc;
    const int tiles_x = (cam->resx + TILE - 1) / TILE;
    if (int rc = ensure_owned_tiles(s, cam->resx, cam->resy, rank, world)) return rc;
    const int n_owned = s->n_owned;
    const int *owned = world == 1 ? nullptr : s->owned_tiles;
    const bool pt = cam->path_tracing != 0;
    // queue growth bound per camera sample: splitting fans out at depth 0, dielectrics may double per bounce
    unsigned growth = (unsigned)(pt && cam->splitting_factor > 1 ? cam->splitting_factor : 1) * (s->has_dielectric ? 2u : 1u);
    // room for the whole shard in one batch (plus the slack the batch planner keeps), as far as the bound allows
    // Identical shadow rays of a vertex are traced once and sphere emitters are evaluated after their ray.  With the
    // work counters on, every light sample is evaluated and traced on its own so that the counts are those of the
    // reference's loop (src/ray_trace.cpp:182-241).
    const char *share_env = getenv("RT_SHARE_SHADOW_RAYS");   // "0": as with the counters on (for A/B measurements)
    const int share_rays = (!(share_env && atoi(share_env) == 0) && !s->counters) ? 1 : 0;
    const unsigned slots_per_vertex = (unsigned)(share_rays ? s->k_pad_deferred : s->k_pad);   // shadow slots of a vertex in one pass
    const int slot_table = share_rays ? s->slot_table : -1;
    // With one sphere emitter as the only light, the shading kernel writes the emitter's one shadow ray per vertex itself
    // (RT_FUSED_EMITTER_RAYS=0: through light_setup_kernel as for any other scene, for A/B timing).
    static const bool fuse_allowed = !(getenv("RT_FUSED_EMITTER_RAYS") && atoi(getenv("RT_FUSED_EMITTER_RAYS")) == 0);
    const int emitter_rays = share_rays && s->single_sphere_emitter && fuse_allowed ? 1 : 0;
    if (int rc = ensure_wavefront(s, (unsigned long long)std::max(n_owned, 1) * 1024ull * n_samples * growth * 4 / 3 + 1024, slots_per_vertex,
                                  1024ull * n_samples * growth * slots_per_vertex * 4 / 3 + 1024, std::max(1, (int)ceilf(s->dev.max_depth) + 2), !emitter_rays))
        return rc;
    const Wavefront &W = s->wf;
    if ((unsigned long long)n_samples * 1024ull > W.capacity) {
        g_error = "NumSamples too large for the wavefront capacity (set RT_WAVEFRONT_CAPACITY)";
        return 5;
    }

    HIP_OK(hipMemsetAsync(s->stats, 0, STAT_ROWS * STAT_WORDS * sizeof(unsigned long long), s->stream));
    HIP_OK(hipEventRecord(s->ev0, s->stream));

    // path depths 0 .. ceil(MaxRecursionDepth); recursive ray tracing of a scene without reflecting or refracting materials
    // ends with the camera rays (src/ray_trace.cpp:681-869 spawns rays for mirror, conductor and dielectric only), and the
    // later generations would be launches over empty queues: 30 of the 35 of a 1080p frame of the 70K-triangle scene
    const int n_gen = !pt && s->only_regular_materials ? 1 : (int)ceilf(s->dev.max_depth) + 1;
    // A depth limit far beyond what paths live (the reference takes any MaxRecursionDepth, src/parser.cpp:1253-1258) must
    // not cost its launches: from generation GEN_CHECK on, every GEN_CHECK generations, the host waits for the shading
    // that fed the next generation's queue and reads its 64 counters; an empty queue makes that generation the last (it runs,
    // over nothing, so that the lighting of the one before is closed as after any last generation).  The two events of a
    // generation are made when the loop reaches it.
    constexpr int GEN_CHECK = 32;
    auto generation_events = [&](int gen) -> int {
        while ((int)s->shaded.size() <= gen) {
            hipEvent_t a, b;
            HIP_OK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            s->shaded.push_back(a);
            HIP_OK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            s->lit.push_back(b);
        }
        return 0;
    };
    // The lighting chain of a generation can run on a second stream beside the closest-hit trace and shading of the next
    // one (RT_TWO_STREAMS=1).  It is off by default: once every kernel fills the chip by itself the overlap gains nothing
    // (4K / 64 spp frame 743 vs 740 ms, 1080p / 1 spp frame 1.27 vs 1.24 ms), and on one stream the per-kernel event times
    // are the kernels' own.
    static const bool two_streams = getenv("RT_TWO_STREAMS") != nullptr && atoi(getenv("RT_TWO_STREAMS")) != 0;
    hipStream_t lighting = two_streams ? s->stream2 : s->stream;
    // What does pay is one pair: the gather of generation g - 1 (arithmetic: the emitter's samples) beside the shading of
    // generation g (memory: queue records streamed).  Both wait for the closest-hit trace of g, which has the chip to itself
    // like every trace launch; the gather goes to the second stream, everything else stays on the first.
    // (4K / 64 spp frame of the 1M-triangle scene: 451.6 -> 440.1 ms; the two take 102 ms together instead of 118 in turn.)
    // Path-traced frames only: a slot then receives the sums of both kernels in whichever order they arrive, and only
    // there is the order of a slot's additions free already (recursive ray tracing keeps its frames bit for bit).
    // Batches of at least 4 Mi slots only: a small frame's launches are short, the hand-over between the streams is not,
    // and a repeated small frame is replayed from a graph instead (below).
    // RT_PAIR_GATHER=0: everything in turn on one stream; =2: batches of any size.
    const int pair_mode = getenv("RT_PAIR_GATHER") ? atoi(getenv("RT_PAIR_GATHER")) : 1;   // (read per frame: a test turns it off and on)
    const int n_pass_all = (s->k_total + 63) / 64;
    const bool pair_possible = pair_mode != 0 && pt && !two_streams && n_gen > 1 && (emitter_rays || n_pass_all == 1);
    const unsigned grid_cap = 4096;
    int tile = 0;
    while (tile < n_owned) {
        // shards fill unevenly once paths branch or die: plan for 3/4 of the room
        unsigned long long slots_max = std::min<unsigned long long>(W.capacity, (unsigned long long)W.shadow_shard_entries * N_SHARDS / slots_per_vertex) / growth;
        if (growth > 1 || pt) slots_max = slots_max * 3 / 4;
        long long per_tile = 1024ll * n_samples;
        // (a multiple of 16 tiles = of 64 workgroups per sample: every shard then gets exactly its share; with any other
        // count a shard gets at most one workgroup more per sample, which a batch below the multiple still has room for)
        long long most = std::max<long long>(1, (long long)slots_max / per_tile);
        if (most >= 32) most &= ~15ll;
        int tiles_now = (int)std::min<long long>(n_owned - tile, most);
        // the camera paths of the batch must fit the fullest shard
        while (tiles_now > 1 && max_shard_groups(4u * (unsigned)tiles_now, (unsigned)n_samples) * 256ull > W.shard_entries)
            tiles_now = tiles_now > 16 ? ((tiles_now - 1) & ~15) : tiles_now - 1;
        if (getenv("RT_PRINT_BATCHES"))   // diagnostic
            fprintf(stderr, "batch: %d tiles of %d left (a batch holds %lld; %u slots per vertex, growth %u)\n", tiles_now, n_owned - tile, most, slots_per_vertex, growth);
        Batch b;
        b.tile_first = tile;
        b.tile_count = tiles_now;
        b.n_pixels = tiles_now * 1024;
        b.n_samples = n_samples;
        b.grid_n = grid_n;
        b.tiles_x = tiles_x;
        b.owned = owned;
        b.still = 0;
        const unsigned slots = (unsigned)b.n_pixels * (unsigned)n_samples;
        const bool pair_gather = pair_possible && (pair_mode == 2 || slots >= (1u << 22));
        // everything the batch puts on the stream(s), in order
        auto issue_batch = [&]() -> int {
            HIP_OK(hipMemsetAsync(W.counters, 0, counter_words(W.generations) * sizeof(unsigned), s->stream));
            // the frame's counters as they stand before this batch, should the batch have to be done again
            HIP_OK(hipMemcpyAsync(s->stats + STAT_ROWS * STAT_WORDS, s->stats, STAT_ROWS * STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s->stream));
            {
                LaunchTimer timer(s, RT_KERNEL_GENERATE, s->stream);
                hipLaunchKernelGGL(generate_kernel, dim3((slots + 255) / 256), dim3(256), 0, s->stream, *cam, b, W, s->stats);
            }
            // consumer grids: a whole number of workgroups per shard
            const unsigned per_shard = (unsigned)std::min<unsigned long long>(grid_cap / N_SHARDS, std::max<unsigned long long>(1, ((unsigned long long)slots * growth / N_SHARDS + 255) / 256));
            const unsigned grid = per_shard * N_SHARDS;
            // `stream` walks the paths (closest hit, shade); `lighting` (the same stream unless RT_TWO_STREAMS is set) lights
            // the vertices of a generation, with two streams while `stream` already traces the next one.  Vertex and shadow
            // buffers alternate by generation parity: shade(g) waits for the lighting of g - 2.
            // the gather of generation `gen` (its only pass, when it is launched apart from the rest of the lighting)
            auto launch_gather = [&](hipStream_t on, int gen) {
                LaunchTimer timer(s, RT_KERNEL_LIGHT_GATHER, on);
                if (emitter_rays)
                    hipLaunchKernelGGL(pt ? light_gather_kernel<true> : light_gather_kernel<false>, dim3(grid), dim3(256), 0, on, s->dev, W, gen, 0, s->light_of, s->k_total, (int)slots_per_vertex, slot_table, 1, pt ? 3 : 2, s->stats);
                else
                    hipLaunchKernelGGL(light_gather_kernel<false>, dim3(grid), dim3(256), 0, on, s->dev, W, gen, 0, s->light_of, s->k_total, (int)slots_per_vertex, slot_table, 1, share_rays && s->pass_has_emitter[0] ? 1 : 0, s->stats);
            };
            int gens = n_gen;   // generations this batch runs: n_gen, or fewer when its paths have all ended
            for (int gen = 0; gen < gens; ++gen) {
                if (int rc = generation_events(gen)) return rc;
                if (gen >= GEN_CHECK && gen % GEN_CHECK == 0 && gen < gens - 1) {   // (never inside a graph capture: see `key` below)
                    unsigned fill[N_SHARDS * COUNTER_STRIDE];
                    HIP_OK(hipStreamSynchronize(s->stream));
                    HIP_OK(hipMemcpy(fill, counter_of(W, Q_PATH, gen, 0), sizeof fill, hipMemcpyDeviceToHost));
                    unsigned long long alive = 0;
                    for (int sh = 0; sh < N_SHARDS; ++sh) alive += fill[sh * COUNTER_STRIDE];
                    if (!alive) gens = gen + 1;
                }
                launch_trace(s, s->stream, false, grid, gen, pt ? 0 : 1);
                if (pair_gather && gen >= 1) {   // the gather of the generation before, beside this generation's shading
                    HIP_OK(hipEventRecord(s->shaded[gen], s->stream));   // (here: "closest hits of `gen` found", and with them the shadow rays of gen - 1)
                    HIP_OK(hipStreamWaitEvent(s->stream2, s->shaded[gen], 0));
                    launch_gather(s->stream2, gen - 1);
                    HIP_OK(hipEventRecord(s->lit[gen - 1], s->stream2));
                }
                if (gen >= 2) HIP_OK(hipStreamWaitEvent(s->stream, s->lit[gen - 2], 0));
                {
                    LaunchTimer timer(s, RT_KERNEL_SHADE, s->stream);
                    typedef void (*ShadeFn)(DevScene, rt_camera_desc, Batch, Wavefront, int, int, unsigned long long *);
                    const ShadeFn fn = s->plain_shading ? (pt ? shade_kernel<true, true> : shade_kernel<false, true>) : (pt ? shade_kernel<true, false> : shade_kernel<false, false>);
                    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, s->stream, s->dev, *cam, b, W, gen, emitter_rays, s->stats);
                }
                const int n_pass = n_pass_all;
                typedef void (*SetupFn)(DevScene, Wavefront, int, int, const int *, int, int, int, int, unsigned long long *);
                auto setup_fn = [&](int pass) -> SetupFn {   // a pass without sphere-emitter samples: the build without their code
                    return (size_t)pass < s->pass_has_emitter.size() && s->pass_has_emitter[(size_t)pass] ? light_setup_kernel<true> : light_setup_kernel<false>;
                };
                if (pair_gather) {   // rays and any-hit trace in turn; the gather follows the next closest-hit trace (the last one: at once)
                    if (!emitter_rays) {
                        LaunchTimer timer(s, RT_KERNEL_LIGHT_SETUP, s->stream);
                        hipLaunchKernelGGL(setup_fn(0), dim3(grid), dim3(256), 0, s->stream, s->dev, W, gen, 0, s->light_of, s->k_total, (int)slots_per_vertex, slot_table, share_rays, s->stats);
                    }
                    launch_trace(s, s->stream, true, grid, gen, emitter_rays ? 1 : 0);
                    if (gen == gens - 1) {
                        launch_gather(s->stream, gen);
                        HIP_OK(hipEventRecord(s->lit[gen], s->stream));
                    }
                    continue;
                }
                HIP_OK(hipEventRecord(s->shaded[gen], s->stream));
                HIP_OK(hipStreamWaitEvent(lighting, s->shaded[gen], 0));
                if (emitter_rays) {   // the rays are there already: trace them, then evaluate the emitter where they arrived
                    launch_trace(s, lighting, true, grid, gen, 1);
                    LaunchTimer timer(s, RT_KERNEL_LIGHT_GATHER, lighting);
                    hipLaunchKernelGGL(pt ? light_gather_kernel<true> : light_gather_kernel<false>, dim3(grid), dim3(256), 0, lighting, s->dev, W, gen, 0, s->light_of, s->k_total, (int)slots_per_vertex, slot_table, 1, pt ? 3 : 2, s->stats);
                }
                for (int pass = 0; pass < n_pass && !emitter_rays; ++pass) {
                    if (pass > 0)
                        HIP_OK(hipMemsetAsync(W.counters + (size_t)(Q_SHADOW * W.generations + gen) * N_SHARDS * COUNTER_STRIDE, 0,
                                              N_SHARDS * COUNTER_STRIDE * sizeof(unsigned), lighting));
                    {
                        LaunchTimer timer(s, RT_KERNEL_LIGHT_SETUP, lighting);
                        hipLaunchKernelGGL(setup_fn(pass), dim3(grid), dim3(256), 0, lighting, s->dev, W, gen, pass, s->light_of, s->k_total, (int)slots_per_vertex, slot_table, share_rays, s->stats);
                    }
                    launch_trace(s, lighting, true, grid, gen, 0);
                    LaunchTimer timer(s, RT_KERNEL_LIGHT_GATHER, lighting);
                    hipLaunchKernelGGL(light_gather_kernel<false>, dim3(grid), dim3(256), 0, lighting, s->dev, W, gen, pass, s->light_of, s->k_total, (int)slots_per_vertex, slot_table, 1, share_rays && s->pass_has_emitter[(size_t)pass] ? 1 : 0, s->stats);
                }
                if (n_pass == 0 && !pt) {   // ambient only
                    LaunchTimer timer(s, RT_KERNEL_LIGHT_GATHER, lighting);
                    hipLaunchKernelGGL(light_gather_kernel<false>, dim3(grid), dim3(256), 0, lighting, s->dev, W, gen, 0, s->light_of, 0, 1, -1, 0, 0, s->stats);
                }
                HIP_OK(hipEventRecord(s->lit[gen], lighting));
            }
            for (int gen = std::max(0, gens - 2); gen < gens; ++gen) HIP_OK(hipStreamWaitEvent(s->stream, s->lit[gen], 0));
            {
                LaunchTimer timer(s, RT_KERNEL_RESOLVE, s->stream);
                hipLaunchKernelGGL(resolve_kernel, dim3((b.n_pixels + 255) / 256), dim3(256), 0, s->stream, *cam, b, W, dev_out, packed ? 1 : 0);
            }
            HIP_OK(hipGetLastError());
            HIP_OK(hipMemcpyAsync(s->host_flags, W.overflow, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
            return 0;
        };
        // A frame of one batch that is asked for a second time with the same camera, buffers and state is captured into a
        // HIP graph and replayed from then on: its forty-odd short launches become one (RT_FRAME_GRAPH=0: never).
        static const bool graphs_allowed = !(getenv("RT_FRAME_GRAPH") && atoi(getenv("RT_FRAME_GRAPH")) == 0);
        unsigned long long key = 0;
        if (graphs_allowed && n_gen <= GEN_CHECK && !s->frame_graph_off && world == 1 && !s->profiling && !s->counters && !two_streams && !pair_gather && tile == 0 && tiles_now == n_owned &&
            !getenv("RT_PRINT_BATCHES")) {
            key = 1469598103934665603ull;
            auto mix = [&](const void *p, size_t n) {
                for (size_t i = 0; i < n; ++i) key = (key ^ ((const unsigned char *)p)[i]) * 1099511628211ull;
            };
            const unsigned long long words[] = {(unsigned long long)rank, (unsigned long long)world, (unsigned long long)packed, (unsigned long long)(uintptr_t)dev_out,
                                                (unsigned long long)(uintptr_t)W.counters, W.capacity, W.shadow_shard_entries, (unsigned long long)emitter_rays,
                                                (unsigned long long)share_rays, growth, (unsigned long long)n_gen, slots_per_vertex, (unsigned long long)tiles_now,
                                                (unsigned long long)(uintptr_t)owned, (unsigned long long)(uintptr_t)s->dev.wnodes, (unsigned long long)W.generations};
            mix(cam, sizeof *cam);
            mix(words, sizeof words);
            key |= 1ull;   // (0 means "none")
        }
        if (key && s->frame_graph_exec && s->frame_graph_key == key) {
            HIP_OK(hipGraphLaunch(s->frame_graph_exec, s->stream));
        } else if (key && s->frame_key_seen == key) {   // the second time: capture, then run what was captured
            drop_frame_graph(s);
            bool captured = hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (captured) {
                const int rc = issue_batch();
                captured = hipStreamEndCapture(s->stream, &s->frame_graph) == hipSuccess && rc == 0 && s->frame_graph &&
                           hipGraphInstantiate(&s->frame_graph_exec, s->frame_graph, nullptr, nullptr, 0) == hipSuccess;
            }
            if (captured) {
                s->frame_graph_key = key;
                HIP_OK(hipGraphLaunch(s->frame_graph_exec, s->stream));
            } else {
                // A capture can be broken from outside (another thread's synchronous copy on the device's default stream is an
                // implicit dependency on the capturing stream).  Nothing has run: the batch goes the ordinary way, and this
                // scene stops trying.
                drop_frame_graph(s);
                (void)hipGetLastError();
                s->frame_graph_off = true;
                s->frame_key_seen = 0;
                if (int rc = issue_batch()) return rc;
            }
        } else {
            s->frame_key_seen = key;
            if (int rc = issue_batch()) return rc;
        }
        HIP_OK(hipStreamSynchronize(s->stream));
        harvest_timers(s);
        if (getenv("RT_PRINT_BATCHES")) {   // diagnostic: how evenly the shards were filled, generation by generation
            std::vector<unsigned> words(counter_words(W.generations));
            if (hipMemcpy(words.data(), W.counters, words.size() * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess)
                for (int g = 0; g < n_gen; ++g) {
                    unsigned long long sum = 0, most = 0;
                    for (int sh = 0; sh < N_SHARDS; ++sh) {
                        const unsigned v = words[((size_t)(Q_PATH * W.generations + g) * N_SHARDS + sh) * COUNTER_STRIDE];
                        sum += v, most = std::max<unsigned long long>(most, v);
                    }
                    if (sum) fprintf(stderr, "  generation %d: %llu paths, fullest shard / mean %.4f\n", g, sum, (double)most * N_SHARDS / (double)sum);
                }
        }
        if (*s->host_flags) {   // a queue ran out of room: this batch again, with fewer camera samples in flight
            drop_frame_graph(s);
            s->frame_key_seen = 0;
            if (tiles_now == 1 && growth > 4096) {
                g_error = "path queue overflow: the scene branches more than the wavefront capacity allows (set RT_WAVEFRONT_CAPACITY)";
                return 6;
            }
            growth *= 2;
            // a small frame got a small wavefront state: let it hold at least one tile at the new bound
            if (int rc = ensure_wavefront(s, 1024ull * n_samples * growth * 4 / 3 + 1024, slots_per_vertex, 1024ull * n_samples * growth * slots_per_vertex * 4 / 3 + 1024,
                                          std::max(1, (int)ceilf(s->dev.max_depth) + 2), !emitter_rays))
                return rc;
            // only this batch is done again (its tiles are simply resolved a second time); the abandoned attempt's counts go
            HIP_OK(hipMemcpyAsync(s->stats, s->stats + STAT_ROWS * STAT_WORDS, STAT_ROWS * STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s->stream));
            continue;
        }
        tile += tiles_now;
    }
    HIP_OK(hipEventRecord(s->ev1, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    kernel_ms = 0.f;
    HIP_OK(hipEventElapsedTime(&kernel_ms, s->ev0, s->ev1));
    return 0;
}

// A device buffer of the scene that holds at least `n_floats` floats (grown when needed, kept).
static int scene_buffer(rt_scene *s, float *&buf, size_t &room, size_t n_floats) {
    if (room >= n_floats && buf) return 0;
    HIP_OK(hipSetDevice(s->device));
    HIP_OK(hipStreamSynchronize(s->stream));
    device_blocks().free(buf);
    buf = nullptr;
    room = 0;
    if (!(buf = (float *)device_blocks().alloc(std::max<size_t>(n_floats, 1) * sizeof(float)))) {
        g_error = "out of device memory for the frame";
        return 10;
    }
    room = n_floats;
    return 0;
}

static bool on_device(const void *p) {
    hipPointerAttribute_t attr;
    const bool dev = hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    return dev;
}

extern "C" {

int rt_render_shard(rt_scene *s, const rt_camera_desc *cam, int32_t rank, int32_t world, float *hdr_out, rt_stats *stats) {
    if (!s || !cam || !hdr_out) {
        g_error = "rt_render: null argument";
        return 1;
    }
    if (world < 1 || rank < 0 || rank >= world) {
        g_error = "rt_render_shard: rank/world out of range";
        return 1;
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    HIP_OK(hipSetDevice(s->device));
    // output: device pointer -> write in place; host pointer -> internal framebuffer + copy back
    const bool out_on_device = on_device(hdr_out);
    const size_t n_floats = (size_t)std::max(cam->resx, 0) * (size_t)std::max(cam->resy, 0) * 3;
    float *dev_out = hdr_out;
    if (!out_on_device && n_floats) {
        if (int rc = scene_buffer(s, s->framebuffer, s->framebuffer_floats, n_floats)) return rc;
        dev_out = s->framebuffer;
        if (world > 1) HIP_OK(hipMemcpyAsync(dev_out, hdr_out, n_floats * sizeof(float), hipMemcpyHostToDevice, s->stream));
    }
    float kernel_ms = 0.f;
    if (int rc = render_tiles(s, cam, rank, world, dev_out, false, kernel_ms)) return rc;
    if (!out_on_device) HIP_OK(hipMemcpy(hdr_out, dev_out, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    auto t1 = std::chrono::high_resolution_clock::now();
    read_stats(s, stats, kernel_ms, (float)std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}

int rt_render(rt_scene *s, const rt_camera_desc *cam, float *hdr_out, rt_stats *stats) { return rt_render_shard(s, cam, 0, 1, hdr_out, stats); }

int32_t rt_queue_shard_of(uint32_t group_of_sample, uint32_t sample) { return queue_shard_of(group_of_sample, sample); }
uint64_t rt_queue_fill_bound(uint32_t groups_per_sample, uint32_t samples) { return max_shard_groups(groups_per_sample, samples); }
int32_t rt_scene_shadow_order_free(const rt_scene *s) { return s && s->shadow_order_free ? 1 : 0; }
int32_t rt_scene_plain_shading(const rt_scene *s) { return s && s->plain_shading ? 1 : 0; }
int32_t rt_scene_wide_records(const rt_scene *s) { return s && s->dev.wnodes ? 1 : 0; }

int64_t rt_selftest_wide_records(const rt_bvh_node *nodes, int32_t n_nodes, int32_t stack_entries) {
    if (!nodes || n_nodes < 3 || nodes[0].n_prims != 0) return -1;
    for (int i = 0; i < n_nodes; ++i)   // (the kernels follow these indices)
        if (nodes[i].n_prims == 0 && !(nodes[i].offset > i && nodes[i].offset < n_nodes && i + 1 < n_nodes && nodes[i].axis >= 0 && nodes[i].axis < 3)) return -2;
    rt_bvh_node *dev = nullptr;
    if (upload<rt_bvh_node>(nodes, (size_t)n_nodes, &dev)) return -2;
    const std::vector<int> top = wide_top_of(nodes);
    DeviceBvh built;
    const int rc = wide_records_of_host_tree(dev, n_nodes, top.data(), (int)top.size(), stack_entries, built, g_error);
    device_blocks().free(dev);
    if (rc) return -2;
    if (!built.wide_nodes) return -1;
    device_blocks().free(built.wide_nodes);
    return built.n_wide_heads;
}
int32_t rt_shadow_order_free(const rt_scene_desc *d) { return d && d->abi_version == RT_ABI_VERSION && shadow_answers_order_free(d) ? 1 : 0; }

// Host-only: which rank renders which 32x32 tile (shard_owner); a rank's list of tiles is the tiles it owns in row-major
// order, which is the order of its packed buffer.
int rt_shard_owner_map(int32_t resx, int32_t resy, int32_t world, int32_t *owner_out) {
    if (resx < 1 || resy < 1 || world < 1 || !owner_out) {
        g_error = "rt_shard_owner_map: bad argument";
        return 1;
    }
    const int tiles_x = (resx + TILE - 1) / TILE, tiles_y = (resy + TILE - 1) / TILE;
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) owner_out[ty * tiles_x + tx] = shard_owner(tx, ty, world);
    return 0;
}

int rt_trace_primary(rt_scene *s, const rt_camera_desc *cam, int32_t *shape_out, int32_t *id_out, float *t_out, rt_stats *stats) {
    if (!s || !cam || !shape_out || !id_out || !t_out) {
        g_error = "rt_trace_primary: null argument";
        return 1;
    }
    int grid_n, n_samples;
    if (int rc = check_camera(s, cam, grid_n, n_samples)) return rc;
    HIP_OK(hipSetDevice(s->device));
    if (int rc = account_paths(s, (unsigned long long)cam->resx * cam->resy)) return rc;
    const int tiles_x = (cam->resx + TILE - 1) / TILE, tiles_y = (cam->resy + TILE - 1) / TILE;
    const int total_tiles = tiles_x * tiles_y;
    if (int rc = ensure_wavefront(s, (unsigned long long)total_tiles * 1024ull, 1, 0, 2)) return rc;
    const Wavefront &W = s->wf;
    const size_t n = (size_t)cam->resx * cam->resy;
    int32_t *d_shape = nullptr, *d_id = nullptr;
    float *d_t = nullptr;
    struct Guard {   // frees the read-out buffers on every return path
        void **p[3];
        ~Guard() {
            for (auto q : p)
                if (*q) (void)hipFree(*q);
        }
    } guard{{(void **)&d_shape, (void **)&d_id, (void **)&d_t}};
    HIP_OK(hipMalloc((void **)&d_shape, n * 4));
    HIP_OK(hipMalloc((void **)&d_id, n * 4));
    HIP_OK(hipMalloc((void **)&d_t, n * 4));
    HIP_OK(hipMemsetAsync(s->stats, 0, STAT_ROWS * STAT_WORDS * sizeof(unsigned long long), s->stream));
    HIP_OK(hipEventRecord(s->ev0, s->stream));
    HIP_OK(hipMemsetAsync(W.overflow, 0, sizeof(unsigned), s->stream));
    // the same generate -> trace stages as a render, first sample of every pixel only
    const int tiles_max = (int)std::max(1u, W.capacity / 1024u);
    for (int tile = 0; tile < total_tiles; tile += tiles_max) {
        Batch b;
        b.tile_first = tile;
        b.tile_count = std::min(tiles_max, total_tiles - tile);
        b.n_pixels = b.tile_count * 1024;
        b.n_samples = 1;
        b.grid_n = grid_n;
        b.tiles_x = tiles_x;
        b.owned = nullptr;
        b.still = 1;
        const unsigned slots = (unsigned)b.n_pixels;
        HIP_OK(hipMemsetAsync(W.counters, 0, (counter_words(W.generations) - 1) * sizeof(unsigned), s->stream));   // all but the overflow flag
        hipLaunchKernelGGL(generate_kernel, dim3((slots + 255) / 256), dim3(256), 0, s->stream, *cam, b, W, s->stats);
        const unsigned grid = std::min(4096u / N_SHARDS, std::max(1u, (slots / N_SHARDS + 255) / 256)) * N_SHARDS;
        launch_trace(s, s->stream, false, grid, 0, 0);
        hipLaunchKernelGGL(primary_out_kernel, dim3(grid), dim3(256), 0, s->stream, s->dev, *cam, b, W, d_shape, d_id, d_t);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(s->ev1, s->stream));
    HIP_OK(hipMemcpyAsync(s->host_flags, W.overflow, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipMemcpyAsync(shape_out, d_shape, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipMemcpyAsync(id_out, d_id, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipMemcpyAsync(t_out, d_t, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    if (*s->host_flags) {
        g_error = "rt_trace_primary: ray queue overflow";
        return 6;
    }
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

#include "rt_tonemap.h"
#include "rt_gather.h"
