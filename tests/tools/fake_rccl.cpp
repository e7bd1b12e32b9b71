// TEST INFRASTRUCTURE -- a stand-in for librccl.so that moves data between THREADS OF ONE PROCESS with device-to-device
// copies, so that the product's RCCL branch (rt_render_gather: status all-reduce, grouped ncclSend / ncclRecv) runs with
// several ranks on a box that has one GPU, which a real communicator refuses (one rank per device).  The library opens it
// through RT_RCCL_LIB=<this .so>.  Only what rt_gather.h calls is implemented, with the semantics it relies on:
//   * ranks join a communicator by (unique id, rank); any number of ranks may share a device
//   * ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd: matched by (source, destination) in posting order; the copy
//     runs on the receiver's stream after everything the sender's stream held when it posted, and the sender's stream
//     continues only after the copy
//   * ncclAllReduce of int32 (sum, max, min) and float32 (sum), in place or not: values meet on the host
// Not part of the product: nothing under raytracer_amd/ or include/ names this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct Post {   // a send waiting for its receive
    const void *src;
    size_t bytes;
    int src_device;
    hipEvent_t ready;      // recorded on the sender's stream when it posted
    hipEvent_t copied;     // recorded on the receiver's stream after the copy
    bool taken = false, done = false;
};

struct World {
    std::mutex m;
    std::condition_variable cv;
    int size = 0;
    int joined = 0;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Post>>> box;   // (source, destination) -> posts in order
    // all-reduce meeting point
    int ar_arrived = 0;
    unsigned long long ar_round = 0;
    std::vector<double> ar_acc;
    std::vector<double> ar_result;
};

std::mutex g_worlds_m;
std::map<std::string, std::shared_ptr<World>> g_worlds;
std::atomic<unsigned long long> g_next_id{1};

struct Comm {
    std::shared_ptr<World> w;
    int rank = 0, device = 0;
};

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *c;
    hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t size_of(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 4;
    }
}

ncclResult_t run_ops() {
    std::vector<Op> ops;
    ops.swap(t_ops);
    // sends first: post and go on; then the receives; then wait until every send has been copied
    std::vector<std::shared_ptr<Post>> mine;
    for (const Op &o : ops) {
        if (!o.send) continue;
        auto p = std::make_shared<Post>();
        p->src = o.buf, p->bytes = o.bytes, p->src_device = o.c->device;
        if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->copied, hipEventDisableTiming) != hipSuccess)
            return ncclUnhandledCudaError;
        if (hipEventRecord(p->ready, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(o.c->w->m);
            o.c->w->box[{o.c->rank, o.peer}].push_back(p);
        }
        o.c->w->cv.notify_all();
        mine.push_back(p);
    }
    for (const Op &o : ops) {
        if (o.send) continue;
        World &w = *o.c->w;
        std::shared_ptr<Post> p;
        {
            std::unique_lock<std::mutex> lk(w.m);
            auto &q = w.box[{o.peer, o.c->rank}];
            w.cv.wait(lk, [&] { return !q.empty(); });
            p = q.front();
            q.pop_front();
        }
        if (p->bytes != o.bytes) return ncclInvalidArgument;   // (the real library leaves a size mismatch undefined)
        if (hipStreamWaitEvent(o.stream, p->ready, 0) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpyAsync(o.buf, p->src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(p->copied, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(w.m);
            p->done = true;
        }
        w.cv.notify_all();
    }
    size_t k = 0;
    for (const Op &o : ops) {
        if (!o.send) continue;
        std::shared_ptr<Post> p = mine[k++];
        World &w = *o.c->w;
        {
            std::unique_lock<std::mutex> lk(w.m);
            w.cv.wait(lk, [&] { return p->done; });
        }
        if (hipStreamWaitEvent(o.stream, p->copied, 0) != hipSuccess) return ncclUnhandledCudaError;
        // the events live as long as the post: released when both sides have dropped it, after the stream work that
        // names them has been queued (HIP keeps an event alive until work that waits on it has run)
        (void)hipEventDestroy(p->ready);
        (void)hipEventDestroy(p->copied);
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    const unsigned long long v = g_next_id++;
    snprintf(id->internal, sizeof id->internal, "fake-rccl-%llu", v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int n, ncclUniqueId id, int rank) {
    if (!out || n < 1 || rank < 0 || rank >= n) return ncclInvalidArgument;
    const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lk(g_worlds_m);
        auto &slot = g_worlds[key];
        if (!slot) slot = std::make_shared<World>(), slot->size = n;
        w = slot;
    }
    if (w->size != n) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->w = w, c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::unique_lock<std::mutex> lk(w->m);   // like the real call: returns when every rank has joined
        ++w->joined;
        w->cv.notify_all();
        w->cv.wait(lk, [&] { return w->joined >= n; });
    }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *out, int n, const int *devices) {
    if (!out || n < 1) return ncclInvalidArgument;
    auto w = std::make_shared<World>();
    w->size = n, w->joined = n;
    for (int r = 0; r < n; ++r) {
        Comm *c = new Comm();
        c->w = w, c->rank = r, c->device = devices ? devices[r] : r;
        out[r] = (ncclComm_t)c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete (Comm *)comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++t_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return run_ops();
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    if (!c || peer < 0 || peer >= c->w->size || peer == c->rank) return ncclInvalidArgument;
    t_ops.push_back(Op{true, (void *)buf, count * size_of(type), peer, c, stream});
    return t_depth > 0 ? ncclSuccess : run_ops();
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    if (!c || peer < 0 || peer >= c->w->size || peer == c->rank) return ncclInvalidArgument;
    t_ops.push_back(Op{false, buf, count * size_of(type), peer, c, stream});
    return t_depth > 0 ? ncclSuccess : run_ops();
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    if (!c || (type != ncclInt32 && type != ncclFloat32) || (op != ncclSum && op != ncclMax && op != ncclMin)) return ncclInvalidArgument;
    std::vector<unsigned char> host(count * 4);
    if (hipMemcpyAsync(host.data(), send, host.size(), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return ncclUnhandledCudaError;
    std::vector<double> v(count);
    for (size_t i = 0; i < count; ++i) v[i] = type == ncclInt32 ? (double)((const int *)host.data())[i] : (double)((const float *)host.data())[i];
    World &w = *c->w;
    std::vector<double> result;
    {
        std::unique_lock<std::mutex> lk(w.m);
        const unsigned long long round = w.ar_round;
        if (w.ar_arrived == 0) w.ar_acc = v;
        else
            for (size_t i = 0; i < count && i < w.ar_acc.size(); ++i)
                w.ar_acc[i] = op == ncclSum ? w.ar_acc[i] + v[i] : (op == ncclMax ? std::max(w.ar_acc[i], v[i]) : std::min(w.ar_acc[i], v[i]));
        if (++w.ar_arrived == w.size) {
            w.ar_result = w.ar_acc;
            w.ar_arrived = 0;
            ++w.ar_round;
            w.cv.notify_all();
        } else {
            w.cv.wait(lk, [&] { return w.ar_round != round; });
        }
        result = w.ar_result;   // (the next round cannot end before this rank has arrived in it)
    }
    for (size_t i = 0; i < count; ++i) {
        if (type == ncclInt32) ((int *)host.data())[i] = (int)result[i];
        else ((float *)host.data())[i] = (float)result[i];
    }
    if (hipMemcpyAsync(recv, host.data(), host.size(), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "invalid argument (stand-in)" : "error (stand-in)"); }

}  // extern "C"
