// Device memory for the buffers a scene lives in and the scratch of its creation: hipMalloc / hipFree behind a small
// per-device cache of freed blocks.  A directory of frames creates and destroys one scene per file -- some forty buffers of
// the same sizes every few milliseconds -- and the driver's allocation calls were a third of that; a block that comes back
// is handed to the next request of (nearly) its size instead.  The cache is bounded (blocks, bytes, block size; the blocks that
// have waited longest go first), holds only
// blocks whose work has finished (callers synchronise before they free, as hipFree would have), and is per device: a block
// goes back to the cache of the device it was allocated on.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

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
                void *p = it->second.first;
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

    // Takes back a block from alloc(); all work that uses it must have finished.  When the cache is full the blocks that
    // have waited longest make room: what a running sequence of scenes frees is what its next scene asks for.
    void free(void *p) {
        if (!p) return;
        std::vector<void *> evicted;
        bool cached = false;
        {
            std::lock_guard<std::mutex> lk(m_);
            for (auto &entry : devices_) {
                PerDevice &d = entry.second;
                auto it = d.live.find(p);
                if (it == d.live.end()) continue;
                const size_t size = it->second;
                d.live.erase(it);
                if (size > MAX_BLOCK) break;
                while (!d.idle.empty() && (d.idle.size() >= MAX_BLOCKS || d.idle_bytes + size > MAX_BYTES)) {
                    auto oldest = d.idle.begin();
                    for (auto k = d.idle.begin(); k != d.idle.end(); ++k)
                        if (k->second.second < oldest->second.second) oldest = k;
                    d.idle_bytes -= oldest->first;
                    evicted.push_back(oldest->second.first);
                    d.idle.erase(oldest);
                }
                d.idle.emplace(size, std::make_pair(p, ++d.stamp));
                d.idle_bytes += size;
                cached = true;
                break;
            }
        }
        for (void *q : evicted) (void)hipFree(q);
        if (!cached) (void)hipFree(p);   // (too large to keep, or not one of ours)
    }

    // Releases idle blocks of `device` until they hold at most `keep_bytes`.
    void trim(int device, size_t keep_bytes) {
        std::vector<void *> out;
        {
            std::lock_guard<std::mutex> lk(m_);
            PerDevice &d = devices_[device];
            while (!d.idle.empty() && d.idle_bytes > keep_bytes) {
                auto it = std::prev(d.idle.end());
                d.idle_bytes -= it->first;
                out.push_back(it->second.first);
                d.idle.erase(it);
            }
        }
        for (void *b : out) (void)hipFree(b);
    }

    size_t idle_bytes(int device) {
        std::lock_guard<std::mutex> lk(m_);
        return devices_[device].idle_bytes;
    }

private:
    struct PerDevice {
        std::multimap<size_t, std::pair<void *, unsigned long long>> idle;   // size -> (block, when it came back)
        unsigned long long stamp = 0;
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
