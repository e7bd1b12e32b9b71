// Study tool (built and run by hand): L2 simulation of the wide walk's memory accesses, logged on the host by
// tests/studies/wide_walk_study.py log DIR.  One 16-way LRU cache (default 4 MB: one XCD's L2) with 128-byte lines; the
// rays of a batch advance in lock step, `inflight` of them at a time, as the persistent trace kernel's lanes do.  The
// first 127 records are taken to be in LDS.  Record layouts: A as built (top breadth-first, then depth-first),
// B breadth-first throughout, C a node next to its first interior child.
//   g++ -O2 -o /tmp/wide_cache_sim tests/studies/wide_cache_sim.cpp && /tmp/wide_cache_sim DIR A 40000 4194304 0
// arguments: directory, layout, rays in flight, cache bytes, rays (-1 all, 0 closest hit, 1 any hit)
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#include <algorithm>
#include <cstring>
static std::vector<int32_t> load_npy_i32(const char *path) {
    FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(1); }
    unsigned char h[10]; if (fread(h, 1, 10, f) != 10) exit(1); int hl = h[8] | h[9] << 8; fseek(f, 10 + hl, SEEK_SET);
    std::vector<int32_t> v; int32_t buf[65536]; size_t n;
    while ((n = fread(buf, 4, 65536, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f); return v;
}
struct Cache {
    int ways, sets; std::vector<uint64_t> tag; std::vector<uint32_t> age; uint32_t clock = 0; long hits = 0, misses = 0;
    Cache(long bytes, int w) : ways(w), sets((int)(bytes / 128 / w)), tag((size_t)sets * w, ~0ull), age((size_t)sets * w, 0) {}
    void access(uint64_t addr) {
        uint64_t line = addr >> 7; int s = (int)((line * 0x9E3779B97F4A7C15ull >> 20) % sets); uint64_t *t = &tag[(size_t)s * ways]; uint32_t *a = &age[(size_t)s * ways];
        ++clock; int victim = 0;
        for (int i = 0; i < ways; ++i) { if (t[i] == line) { a[i] = clock; ++hits; return; } if (a[i] < a[victim]) victim = i; }
        t[victim] = line; a[victim] = clock; ++misses;
    }
};
int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: wide_cache_sim DIR [layout] [inflight] [cache bytes] [rays]\n"); return 1; }
    auto log = load_npy_i32((std::string(argv[1]) + "/widelog.npy").c_str());
    auto refs = load_npy_i32((std::string(argv[1]) + "/widerefs.npy").c_str());
    --argc, ++argv;
    const int n = (int)refs.size() / 4;
    const char *layout = argc > 1 ? argv[1] : "A";
    int inflight = argc > 2 ? atoi(argv[2]) : 40000;
    long cache_bytes = argc > 3 ? atol(argv[3]) : 4l << 20;
    int only_any = argc > 4 ? atoi(argv[4]) : -1;   // -1 all, 0 closest, 1 any
    std::vector<int64_t> addr(n);
    if (layout[0] == 'A') for (int i = 0; i < n; ++i) addr[i] = (int64_t)i * 64;
    else if (layout[0] == 'B') {   // breadth-first
        std::vector<int> q{0}; int64_t at = 0;
        for (size_t k = 0; k < q.size(); ++k) { int i = q[k]; addr[i] = at; at += 64; for (int c = 0; c < 4; ++c) { int r = refs[4 * i + c]; if (r >= 0 && r < 0x40000000) q.push_back(r); } }
    } else if (layout[0] == 'C') {   // pairs: a node and its first interior child share a line, recursively (depth-first over pairs)
        std::vector<char> placed(n, 0); std::vector<int> st{0}; int64_t at = 0;
        while (!st.empty()) { int i = st.back(); st.pop_back(); if (placed[i]) continue; placed[i] = 1; addr[i] = at; at += 64;
            int first = -1; std::vector<int> kids;
            for (int c = 0; c < 4; ++c) { int r = refs[4 * i + c]; if (r >= 0 && r < 0x40000000) kids.push_back(r); }
            if ((at & 127) && !kids.empty()) { first = kids[0]; placed[first] = 1; addr[first] = at; at += 64; }
            // push grandchildren of `first` and other kids
            std::vector<int> todo;
            for (int k : kids) if (k != first) todo.push_back(k);
            if (first >= 0) for (int c = 0; c < 4; ++c) { int r = refs[4 * first + c]; if (r >= 0 && r < 0x40000000) todo.push_back(r); }
            for (size_t k = todo.size(); k-- > 0;) st.push_back(todo[k]);
        }
    }
    const int64_t prim_base = (int64_t)n * 64 + (1 << 20);   // a leaf reached = its first entry's 64-byte block (box + entry)
    // split into rays
    std::vector<std::pair<size_t, size_t>> rays; size_t b = 0;
    for (size_t i = 0; i < log.size(); ++i) if (log[i] <= -(1 << 30)) { int any = log[i] == -(1 << 30) - 1; if (only_any < 0 || only_any == any) rays.push_back({b, i}); b = i + 1; }
    Cache cache(cache_bytes, 16);
    long node_acc = 0, node_miss = 0, leaf_acc = 0, leaf_miss = 0;
    for (size_t r0 = 0; r0 < rays.size(); r0 += inflight) {
        size_t r1 = std::min(rays.size(), r0 + inflight); bool more = true;
        for (size_t k = 0; more; ++k) { more = false;
            for (size_t r = r0; r < r1; ++r) { size_t i = rays[r].first + k; if (i >= rays[r].second) continue; more = true;
                int e = log[i]; long m0 = cache.misses;
                if (e >= 0) { if (e >= 127) { cache.access(addr[e]); ++node_acc; node_miss += cache.misses - m0; } }
                else { int first = -1 - e; cache.access(prim_base + 64ll * first); leaf_acc += 1; leaf_miss += cache.misses - m0; }
            } } }
    printf("layout %s inflight %d cache %ld MB rays %zu: node accesses/ray %.2f misses/ray %.2f (%.0f%%), leaf accesses/ray %.2f misses/ray %.2f; bytes/ray %.0f\n", layout, inflight, cache_bytes >> 20, rays.size(),
           (double)node_acc / rays.size(), (double)node_miss / rays.size(), 100.0 * node_miss / node_acc, (double)leaf_acc / rays.size(), (double)leaf_miss / rays.size(), 128.0 * (node_miss + leaf_miss) / rays.size());
}
