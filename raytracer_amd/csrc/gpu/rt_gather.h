// One frame on several GPUs: every rank renders its own 32x32 tiles (shard_owner) into a packed, tile-major buffer
// and rank 0 gathers the buffers and writes the frame.  Included by render.hip (shares its error string and helpers).
//
// The reference runs one process with one thread per core and hands 32x32 tiles to the threads through an atomic
// counter (src/main.cpp:437-555, tile size :441); here a rank is a GPU, the hand-out is static, and the one exchange of
// the path is the gather of the finished tiles: one grouped ncclSend / ncclRecv per frame (RCCL; each peer's packed
// buffer travels over its own direct link to the root).  Ranks are threads of one process (rt_comm_init_all, the
// command line) or processes (rt_comm_init_rank with an id passed around by the launcher, bench.py under torchrun).
// Ranks that share a device -- a rehearsal of the N-rank path on one GPU, which RCCL does not allow in one
// communicator -- exchange through device-to-device copies inside the process instead; packing, tile lists and the
// unpack kernel are the same.
//
// A frame fails as a whole: before the tiles travel the ranks agree on one status word (an all-reduce of one integer on
// the RCCL transport, the meeting point's flag on the local one), so a rank whose render failed, or that could not get
// its buffers, still takes part in every step its peers wait in, and every rank of the frame returns an error (its own,
// or 12 for a peer's) instead of some of them waiting for ever.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <memory>

namespace {

// RCCL is loaded when the first communicator is made: a single-GPU process never maps it.
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // first the RCCL that belongs to the HIP runtime this process runs on (the one in the runtime's own directory: a
        // process may hold a second ROCm installation, e.g. the one a Python package ships), then the loader's choice
        // (RT_RCCL_LIB=<path> goes first: a site's own build of the library, or the tests' stand-in that moves the data
        // between the threads of one process so that the exchange below runs with several ranks on one GPU)
        std::vector<std::string> names;
        if (const char *e = getenv("RT_RCCL_LIB"))
            if (*e) names.push_back(e);
        Dl_info info;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1");
                names.push_back(dir + "librccl.so");
            }
        }
        names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
        for (const std::string &name : names) {
            api.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        auto sym = [&](const char *n) { return dlsym(api.handle, n); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommInitAll || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send ||
            !api.Recv || !api.AllReduce || !api.GetErrorString) {
            dlclose(api.handle);
            api.handle = nullptr;
        }
    });
    return api.handle ? &api : nullptr;
}

#define RCCL_OK(api, expr)                                                                   \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            g_error = std::string(#expr) + ": " + (api)->GetErrorString(r_);                 \
            return 11;                                                                       \
        }                                                                                    \
    } while (0)

// Meeting point of ranks that are threads of this process and share a device (the rehearsal transport).
struct LocalGroup {
    std::mutex m;
    std::condition_variable cv;
    int world = 0;
    int arrived = 0, left = 0;
    unsigned long long round = 0;
    std::vector<const float *> packed;   // per rank: its finished packed buffer (device memory)
    int failed = 0;                      // a rank of the round that is meeting has failed
    int last_failed = 0;                 // the finished round's verdict, for the peers that wake up after it
};

}  // namespace

struct rt_comm {
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;               // null: local transport
    std::shared_ptr<LocalGroup> local;
    // the layout of the last gathered frame
    int key[3] = {0, 0, 0};                  // resx, resy, world
    std::vector<int> tiles_of_rank;          // tiles per rank
    std::vector<size_t> first_tile_of_rank;  // prefix sums: where a rank's tiles start in the root's receive buffer
    int *tile_slot = nullptr;                // root, device: frame tile (row-major) -> its slot in the receive buffer
    size_t tile_slot_room = 0;
    int device = 0;
    float *packed = nullptr;                 // this rank's tiles (ranks > 0)
    size_t packed_room = 0;
    float *gathered = nullptr;               // root: every rank's tiles, rank after rank
    size_t gathered_room = 0;
    float *frame = nullptr;                  // root: the frame on the device when the caller's buffer is host memory
    size_t frame_room = 0;
    int *status = nullptr;                   // RCCL transport, device: the frame's status word (made with the communicator)
};

namespace {

// root: the gathered tiles to their places in the frame (row 0 = bottom row, as everywhere)
__global__ void __launch_bounds__(256) unpack_tiles_kernel(const float *__restrict__ gathered, const int *__restrict__ tile_slot, int tiles_x, int resx,
                                                            int resy, float *__restrict__ frame) {
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const float *src = gathered + (size_t)tile_slot[tile] * (TILE * TILE * 3);
    for (int e = threadIdx.x; e < TILE * TILE; e += 256) {
        const int i = tx * TILE + (e % TILE), j = ty * TILE + (e / TILE);
        if (i >= resx || j >= resy) continue;
        float *dst = frame + ((size_t)j * resx + i) * 3;
        dst[0] = src[3 * e], dst[1] = src[3 * e + 1], dst[2] = src[3 * e + 2];
    }
}

int comm_buffer(rt_comm *c, float *&buf, size_t &room, size_t n_floats) {
    if (buf && room >= n_floats) return 0;
    if (buf) (void)hipFree(buf);
    buf = nullptr;
    room = 0;
    HIP_OK(hipMalloc((void **)&buf, std::max<size_t>(n_floats, 1) * sizeof(float)));
    room = n_floats;
    return 0;
}

// The layout of a gathered frame: tiles per rank, and for every frame tile (row-major) its slot in the root's buffer, where
// rank r's tiles follow those of ranks 0 .. r - 1 in row-major order (the order of the rank's own list).
void shard_layout(int resx, int resy, int world, std::vector<int> &tiles_of_rank, std::vector<int> &tile_slot) {
    const int tiles_x = (resx + TILE - 1) / TILE, tiles_y = (resy + TILE - 1) / TILE;
    tiles_of_rank.assign((size_t)world, 0);
    tile_slot.assign((size_t)tiles_x * tiles_y, 0);
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) tile_slot[(size_t)ty * tiles_x + tx] = tiles_of_rank[(size_t)shard_owner(tx, ty, world)]++;
    std::vector<int> first((size_t)world, 0);
    for (int r = 1; r < world; ++r) first[(size_t)r] = first[(size_t)r - 1] + tiles_of_rank[(size_t)r - 1];
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) tile_slot[(size_t)ty * tiles_x + tx] += first[(size_t)shard_owner(tx, ty, world)];
}

int comm_layout(rt_comm *c, int resx, int resy) {
    const int key[3] = {resx, resy, c->world};
    if (!c->tiles_of_rank.empty() && memcmp(key, c->key, sizeof key) == 0) return 0;
    std::vector<int> slot;
    shard_layout(resx, resy, c->world, c->tiles_of_rank, slot);
    c->first_tile_of_rank.assign((size_t)c->world + 1, 0);
    for (int r = 0; r < c->world; ++r) c->first_tile_of_rank[(size_t)r + 1] = c->first_tile_of_rank[(size_t)r] + (size_t)c->tiles_of_rank[(size_t)r];
    if (c->rank == 0) {
        if (c->tile_slot_room < slot.size()) {
            if (c->tile_slot) (void)hipFree(c->tile_slot);
            c->tile_slot = nullptr;
            c->tile_slot_room = 0;
            HIP_OK(hipMalloc((void **)&c->tile_slot, slot.size() * sizeof(int)));
            c->tile_slot_room = slot.size();
        }
        HIP_OK(hipMemcpy(c->tile_slot, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    memcpy(c->key, key, sizeof key);
    return 0;
}

}  // namespace

extern "C" {

int rt_shard_layout(int32_t resx, int32_t resy, int32_t world, int32_t *tiles_of_rank_out, int32_t *tile_slot_out) {
    if (resx < 1 || resy < 1 || world < 1 || !tiles_of_rank_out || !tile_slot_out) {
        g_error = "rt_shard_layout: bad argument";
        return 1;
    }
    std::vector<int> counts, slot;
    shard_layout(resx, resy, world, counts, slot);
    memcpy(tiles_of_rank_out, counts.data(), counts.size() * sizeof(int));
    memcpy(tile_slot_out, slot.data(), slot.size() * sizeof(int));
    return 0;
}

int32_t rt_comm_available(void) { return rccl_api() ? 1 : 0; }

int rt_comm_unique_id(void *id_out, int32_t id_bytes) {
    RcclApi *api = rccl_api();
    if (!api) {
        g_error = "rt_comm_unique_id: librccl.so could not be loaded";
        return 11;
    }
    if (!id_out || id_bytes < (int32_t)sizeof(ncclUniqueId)) {
        g_error = "rt_comm_unique_id: the id needs RT_COMM_ID_BYTES bytes";
        return 1;
    }
    ncclUniqueId id;
    RCCL_OK(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int rt_comm_init_rank(const void *id, int32_t id_bytes, int32_t rank, int32_t world, rt_comm **out) {
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id) || (id && id_bytes < (int32_t)sizeof(ncclUniqueId))) {
        g_error = "rt_comm_init_rank: bad argument";
        return 1;
    }
    std::unique_ptr<rt_comm> c(new rt_comm());
    c->rank = rank;
    c->world = world;
    HIP_OK(hipGetDevice(&c->device));
    if (id) {   // (a single rank may do without an id, and then without RCCL)
        RcclApi *api = rccl_api();
        if (!api) {
            g_error = "rt_comm_init_rank: librccl.so could not be loaded";
            return 11;
        }
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof uid);
        RCCL_OK(api, api->CommInitRank(&c->nccl, world, uid, rank));
        if (hipMalloc((void **)&c->status, 2 * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            (void)api->CommDestroy(c->nccl);
            g_error = "rt_comm_init_rank: no device memory for the status word";
            return 2;
        }
    }
    *out = c.release();
    return 0;
}

int rt_comm_init_all(int32_t n, const int32_t *devices, rt_comm **comms_out) {
    if (n < 1 || !devices || !comms_out) {
        g_error = "rt_comm_init_all: bad argument";
        return 1;
    }
    bool distinct = true;
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b) distinct = distinct && devices[a] != devices[b];
    std::vector<ncclComm_t> nccl((size_t)n, nullptr);
    std::shared_ptr<LocalGroup> group;
    if (n > 1 && distinct) {
        RcclApi *api = rccl_api();
        if (!api) {
            g_error = "rt_comm_init_all: librccl.so could not be loaded";
            return 11;
        }
        std::vector<int> devs(devices, devices + n);
        RCCL_OK(api, api->CommInitAll(nccl.data(), n, devs.data()));
    } else if (n > 1) {   // ranks share a device: the in-process transport
        group = std::make_shared<LocalGroup>();
        group->world = n;
        group->packed.assign((size_t)n, nullptr);
    }
    int before = 0;
    (void)hipGetDevice(&before);
    for (int r = 0; r < n; ++r) {
        rt_comm *c = new rt_comm();
        c->rank = r;
        c->world = n;
        c->device = devices[r];
        c->nccl = nccl[(size_t)r];
        c->local = group;
        comms_out[r] = c;
        if (c->nccl && (hipSetDevice(c->device) != hipSuccess || hipMalloc((void **)&c->status, 2 * sizeof(int)) != hipSuccess)) {
            (void)hipGetLastError();
            for (int k = 0; k <= r; ++k) rt_comm_destroy(comms_out[k]), comms_out[k] = nullptr;
            for (int k = r + 1; k < n; ++k)
                if (nccl[(size_t)k]) (void)rccl_api()->CommDestroy(nccl[(size_t)k]);
            (void)hipSetDevice(before);
            g_error = "rt_comm_init_all: no device memory for the status word";
            return 2;
        }
    }
    (void)hipSetDevice(before);
    return 0;
}

void rt_comm_destroy(rt_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->nccl)
        if (RcclApi *api = rccl_api()) (void)api->CommDestroy(c->nccl);
    for (void *p : {(void *)c->tile_slot, (void *)c->packed, (void *)c->gathered, (void *)c->frame, (void *)c->status})
        if (p) (void)hipFree(p);
    delete c;
}

int rt_render_gather(rt_scene *s, rt_comm *c, const rt_camera_desc *cam, float *hdr_out, rt_stats *stats) {
    if (!s || !c || !cam) {   // (nothing to take part in the exchange with)
        g_error = "rt_render_gather: null argument";
        return 1;
    }
    // From here on every path reaches the point where the ranks agree on the frame's status: `rc` is this rank's own.
    int rc = 0;
    std::string own_error;
    auto failed = [&](int code) {
        if (code && !rc) rc = code, own_error = g_error;
        return code != 0;
    };
    if (c->rank == 0 && !hdr_out) {
        g_error = "rt_render_gather: null argument";
        failed(1);
    } else if (s->device != c->device) {
        g_error = "rt_render_gather: the scene and the communicator are on different devices";
        failed(1);
    } else if (cam->resx < 1 || cam->resy < 1) {
        g_error = "invalid camera: resolution and sample count must be positive";
        failed(4);
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    HIP_OK(hipSetDevice(c->device));
    const size_t tile_floats = (size_t)TILE * TILE * 3;
    size_t mine = 0, all = 0, frame_floats = 0;
    int tiles_x = 0, tiles_y = 0;
    float kernel_ms = 0.f;
    if (!rc && !failed(comm_layout(c, cam->resx, cam->resy))) {
        mine = (size_t)c->tiles_of_rank[(size_t)c->rank] * tile_floats;
        all = c->first_tile_of_rank[(size_t)c->world] * tile_floats;
        frame_floats = (size_t)cam->resx * cam->resy * 3;
        tiles_x = (cam->resx + TILE - 1) / TILE, tiles_y = (cam->resy + TILE - 1) / TILE;
        float *target = nullptr;   // where this rank's tiles are rendered: straight into the root's gather buffer, or the send buffer
        if (c->rank == 0) {
            if (!failed(comm_buffer(c, c->gathered, c->gathered_room, all))) target = c->gathered + c->first_tile_of_rank[0] * tile_floats;
        } else {
            if (!failed(comm_buffer(c, c->packed, c->packed_room, mine))) target = c->packed;
        }
        if (!rc) failed(render_tiles(s, cam, c->rank, c->world, target, true, kernel_ms));   // returns with the stream drained
    }
    auto verdict_of = [&](int any_failed) {   // the frame's verdict as this rank's return value
        if (!any_failed) return 0;
        if (rc) {
            g_error = own_error;
            return rc;
        }
        g_error = "rt_render_gather: a peer rank failed to render its tiles";
        return 12;
    };
    // ---- the exchange: one status word, then every rank's packed tiles to the root ----
    if (c->nccl) {
        RcclApi *api = rccl_api();
        if (c->world > 1) {
            int status = rc ? 1 : 0;
            HIP_OK(hipMemcpyAsync(c->status, &status, sizeof status, hipMemcpyHostToDevice, s->stream));
            RCCL_OK(api, api->AllReduce(c->status, c->status + 1, 1, ncclInt32, ncclMax, c->nccl, s->stream));
            HIP_OK(hipMemcpyAsync(&status, c->status + 1, sizeof status, hipMemcpyDeviceToHost, s->stream));
            HIP_OK(hipStreamSynchronize(s->stream));
            if (status) return verdict_of(1);
        } else if (rc) {
            return verdict_of(1);
        }
        RCCL_OK(api, api->GroupStart());
        ncclResult_t r = ncclSuccess;
        const char *what = "";
        if (c->rank == 0) {
            for (int k = 1; k < c->world && r == ncclSuccess; ++k)
                if (c->tiles_of_rank[(size_t)k]) {
                    r = api->Recv(c->gathered + c->first_tile_of_rank[(size_t)k] * tile_floats, (size_t)c->tiles_of_rank[(size_t)k] * tile_floats, ncclFloat, k, c->nccl,
                                  s->stream);
                    what = "ncclRecv";
                }
        } else if (mine) {
            r = api->Send(c->packed, mine, ncclFloat, 0, c->nccl, s->stream);
            what = "ncclSend";
        }
        const ncclResult_t closed = api->GroupEnd();   // (also after a failed call inside: the group must not stay open on this thread)
        if (r != ncclSuccess || closed != ncclSuccess) {
            g_error = std::string(r != ncclSuccess ? what : "ncclGroupEnd") + ": " + api->GetErrorString(r != ncclSuccess ? r : closed);
            return 11;
        }
    } else if (c->local) {
        LocalGroup &g = *c->local;
        std::unique_lock<std::mutex> lk(g.m);
        const unsigned long long round = g.round;
        g.packed[(size_t)c->rank] = c->rank == 0 || rc ? nullptr : c->packed;
        if (rc) g.failed = 1;
        if (++g.arrived == g.world) g.cv.notify_all();
        int any_failed = 0;
        if (c->rank == 0) {
            g.cv.wait(lk, [&] { return g.arrived == g.world; });
            any_failed = g.failed;
            hipError_t e = hipSuccess;
            for (int r = 1; r < c->world && !any_failed && e == hipSuccess; ++r)
                if (c->tiles_of_rank[(size_t)r])
                    e = hipMemcpyAsync(c->gathered + c->first_tile_of_rank[(size_t)r] * tile_floats, g.packed[(size_t)r],
                                       (size_t)c->tiles_of_rank[(size_t)r] * tile_floats * sizeof(float), hipMemcpyDeviceToDevice, s->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            g.arrived = 0;
            g.failed = 0;
            g.last_failed = any_failed;
            ++g.round;          // the peers' buffers are free again
            g.cv.notify_all();
            lk.unlock();
            if (!any_failed) HIP_OK(e);
        } else {
            g.cv.wait(lk, [&] { return g.round != round; });
            any_failed = g.last_failed;   // (the next round cannot end before this rank has arrived in it)
        }
        if (any_failed) return verdict_of(1);
    } else if (rc) {
        return verdict_of(1);
    }
    // ---- root: tiles to their places in the frame ----
    if (c->rank == 0) {
        const bool out_on_device = on_device(hdr_out);
        float *dev_frame = hdr_out;
        if (!out_on_device) {
            if (int rc2 = comm_buffer(c, c->frame, c->frame_room, frame_floats)) return rc2;
            dev_frame = c->frame;
        }
        hipLaunchKernelGGL(unpack_tiles_kernel, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, s->stream, c->gathered, c->tile_slot, tiles_x, cam->resx,
                           cam->resy, dev_frame);
        HIP_OK(hipGetLastError());
        if (!out_on_device) HIP_OK(hipMemcpyAsync(hdr_out, dev_frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    }
    HIP_OK(hipStreamSynchronize(s->stream));
    auto t1 = std::chrono::high_resolution_clock::now();
    read_stats(s, stats, kernel_ms, (float)std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}

}  // extern "C"
