// Device memory for the buffers a scene lives in and the scratch of its creation: hipMalloc / hipFree behind a small
// per-device cache of freed blocks.  A directory of frames creates and destroys one scene per file -- some forty buffers of
// the same sizes every few milliseconds -- and the driver's allocation calls were a third of that; a block that comes back
// is handed to the next request of (nearly) its size instead.  The cache is bounded (blocks, bytes, block size), holds only
// blocks whose work has finished (callers synchronise before they free, as hipFree would have), and is per device: a block
// goes back to the cache of the device it was allocated on.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <map>
#include <mutex>
#include <unordered_map>

namespace rt {

class DeviceBlockCache {
public:
    static constexpr size_t MAX_BLOCK = 256ull << 20;    // larger blocks are never cached
    static constexpr size_t MAX_BYTES = 1024ull << 20;   // what the idle blocks of one device may hold together
    static constexpr size_t MAX_BLOCKS = 512;

    // A block of at least `bytes` bytes on the current device, or nullptr.
    void *alloc(size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        const size_t want = (bytes < 256 ? 256 : bytes + 255) / 256 * 256;
        {
            std::lock_guard<std::mutex> lk(m_);
            PerDevice &d = devices_[dev];
            auto it = d.idle.lower_bound(want);
            if (it != d.idle.end() && it->first <= want + want / 8 + 4096) {
                void *p = it->second;
                d.idle_bytes -= it->first;
                d.live[p] = it->first;
                d.idle.erase(it);
                return p;
            }
        }
        void *p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            trim(dev, 0);   // give the idle blocks back and try once more
            if (hipMalloc(&p, want) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
        }
        std::lock_guard<std::mutex> lk(m_);
        devices_[dev].live[p] = want;
        return p;
    }

    // Takes back a block from alloc(); all work that uses it must have finished.
    void free(void *p) {
        if (!p) return;
        std::unique_lock<std::mutex> lk(m_);
        for (auto &entry : devices_) {
            PerDevice &d = entry.second;
            auto it = d.live.find(p);
            if (it == d.live.end()) continue;
            const size_t size = it->second;
            d.live.erase(it);
            if (size <= MAX_BLOCK && d.idle.size() < MAX_BLOCKS && d.idle_bytes + size <= MAX_BYTES) {
                d.idle.emplace(size, p);
                d.idle_bytes += size;
                return;
            }
            lk.unlock();
            (void)hipFree(p);
            return;
        }
        lk.unlock();
        (void)hipFree(p);   // (not one of ours)
    }

    // Releases idle blocks of `device` until they hold at most `keep_bytes`.
    void trim(int device, size_t keep_bytes) {
        std::multimap<size_t, void *> out;
        {
            std::lock_guard<std::mutex> lk(m_);
            PerDevice &d = devices_[device];
            while (!d.idle.empty() && d.idle_bytes > keep_bytes) {
                auto it = std::prev(d.idle.end());
                d.idle_bytes -= it->first;
                out.insert(*it);
                d.idle.erase(it);
            }
        }
        for (auto &b : out) (void)hipFree(b.second);
    }

    size_t idle_bytes(int device) {
        std::lock_guard<std::mutex> lk(m_);
        return devices_[device].idle_bytes;
    }

private:
    struct PerDevice {
        std::multimap<size_t, void *> idle;             // size -> block
        std::unordered_map<void *, size_t> live;        // blocks handed out
        size_t idle_bytes = 0;
    };
    std::mutex m_;
    std::map<int, PerDevice> devices_;
};

inline DeviceBlockCache &device_blocks() {
    static DeviceBlockCache cache;
    return cache;
}

}  // namespace rt
