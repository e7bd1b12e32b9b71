// Binary scene cache: the flat arrays a scene file is turned into (triangles, placements, BVH, texels, ...), kept in one
// file so that loading the same scene again is a memory map instead of parsing JSON / PLY / image files and building the
// BVH (reference src/parser.cpp:414-452 parses vertex text through istringstream, src/ply_loader.cpp:185-294 compares
// property names per value, src/bvh.cpp:41-344 builds on one thread).
//
// Enabled by the environment variable RT_SCENE_CACHE=<directory>.  A cache file is named after the scene file's path
// and content; it records every file the scene pulled in (PLY meshes, images) with a 128-bit hash of its content, and
// is used only while all of them still hash the same.  The arrays are mapped read-only and handed to rt_scene_create as
// they lie in the file; the scene description built from a cache file is byte for byte the one a fresh load builds.
// The header also carries a 128-bit hash of everything behind it (checked on every load: a truncated, damaged or edited
// file falls back to the full load) and a stamp of the host sources the library was built from (the build passes their
// hash as RT_HOST_SOURCE_STAMP), so a cache written by another ingest or tree builder is not used either.  The hash is not
// cryptographic: the cache directory is trusted as far as the scene files themselves are.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <memory>
#include <thread>

#include "scene.h"

namespace rth {

namespace {

constexpr uint32_t CACHE_MAGIC = 0x43535452u;   // "RTSC"
constexpr uint32_t CACHE_VERSION = 2;
#ifndef RT_HOST_SOURCE_STAMP
#define RT_HOST_SOURCE_STAMP __DATE__ " " __TIME__   // (a build that does not hash the sources: the time this file was compiled)
#endif
constexpr int N_SECTIONS = 13;

struct Hash128 {
    uint64_t a = 0x9e3779b97f4a7c15ull, b = 0xc2b2ae3d27d4eb4full;
    void words(const uint64_t *w, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            a = (a ^ w[i]) * 0x100000001b3ull;
            a ^= a >> 29;
            b = (b + w[i]) * 0xff51afd7ed558ccdull;
            b ^= b >> 32;
        }
    }
    void bytes(const void *p, size_t n) {
        const uint8_t *s = (const uint8_t *)p;
        const size_t whole = n / 8;
        if (((uintptr_t)s & 7) == 0) {
            words((const uint64_t *)s, whole);
        } else {
            for (size_t i = 0; i < whole; ++i) {
                uint64_t w;
                memcpy(&w, s + 8 * i, 8);
                words(&w, 1);
            }
        }
        uint64_t tail = 0;
        memcpy(&tail, s + 8 * whole, n - 8 * whole);
        tail ^= (uint64_t)n << 56;
        words(&tail, 1);
    }
};

// A file mapped read-only (empty files map to nothing and count as readable).
struct Mapping {
    const uint8_t *data = nullptr;
    size_t size = 0;
    bool ok = false;
    explicit Mapping(const std::string &path, bool populate = false) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            size = (size_t)st.st_size;
            if (size == 0) {
                ok = true;
            } else {
                void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), fd, 0);
                if (p != MAP_FAILED) {
                    data = (const uint8_t *)p;
                    ok = true;
                }
            }
        }
        close(fd);
    }
    ~Mapping() {
        if (data) munmap((void *)data, size);
    }
    Mapping(const Mapping &) = delete;
    Mapping &operator=(const Mapping &) = delete;
};

// Hash of a large block: 1 MiB pieces hashed on up to eight threads, the pieces' hashes folded in order (the result does
// not depend on the number of threads).
void hash_block(const uint8_t *p, size_t n, Hash128 &h) {
    constexpr size_t PIECE = 1u << 20;
    const size_t pieces = (n + PIECE - 1) / PIECE;
    std::vector<Hash128> part(pieces);
    size_t want = std::min<size_t>(8, pieces / 8);   // a thread is worth starting for eight pieces or more
    want = std::min<size_t>(want, std::max(1u, std::thread::hardware_concurrency()));
    const unsigned workers = (unsigned)std::max<size_t>(1, want);
    auto run = [&](unsigned w) {
        for (size_t i = w; i < pieces; i += workers) part[i].bytes(p + i * PIECE, std::min(PIECE, n - i * PIECE));
    };
    std::vector<std::thread> pool;
    for (unsigned w = 1; w < workers; ++w) pool.emplace_back(run, w);
    run(0);
    for (std::thread &t : pool) t.join();
    for (const Hash128 &x : part) {
        const uint64_t v[2] = {x.a, x.b};
        h.words(v, 2);
    }
    const uint64_t len = n;
    h.words(&len, 1);
}

uint64_t source_stamp() {
    Hash128 h;
    const char *s = RT_HOST_SOURCE_STAMP;
    h.bytes(s, strlen(s));
    return h.a ^ h.b;
}

bool hash_file(const std::string &path, Hash128 &h) {
    Mapping m(path);
    if (!m.ok) return false;
    h.bytes(m.data, m.size);
    return true;
}

struct Header {
    uint32_t magic, version, abi_version, n_sections;
    uint64_t json_hash[2];
    uint64_t section_offset[N_SECTIONS], section_bytes[N_SECTIONS];
    uint64_t tail_offset, tail_bytes;   // dependencies, cameras, warnings
    float intersection_epsilon, shadow_ray_epsilon, max_depth;
    float ambient_light[3], background_color[3];
    int32_t background_texture, is_probe;
    uint32_t struct_bytes[N_SECTIONS];   // sizeof of every record type when the file was written
    uint64_t source_stamp;               // of the host sources that wrote the file
    uint64_t body_bytes;                 // the file's size behind the header (64-byte aligned start) ...
    uint64_t body_hash[2];               // ... and its hash
};
constexpr uint64_t BODY_START = (sizeof(Header) + 63) / 64 * 64;

const uint32_t RECORD_BYTES[N_SECTIONS] = {sizeof(rt_triangle), sizeof(rt_instance), sizeof(rt_sphere), sizeof(rt_plane), sizeof(rt_material),
                                           sizeof(rt_texture),  sizeof(int32_t),     sizeof(rt_image),  sizeof(uint8_t),  sizeof(float),
                                           sizeof(rt_light),    sizeof(rt_bvh_node), sizeof(rt_bvh_prim)};

std::string cache_dir() {
    const char *d = getenv("RT_SCENE_CACHE");
    return d && *d ? std::string(d) : std::string();
}

// name of the scene's cache file: hash of its absolute path and its content
bool cache_path_for(const std::string &json_path, std::string &path_out, uint64_t json_hash[2]) {
    const std::string dir = cache_dir();
    if (dir.empty()) return false;
    Hash128 h;
    std::error_code ec;
    const std::string abs = std::filesystem::absolute(json_path, ec).lexically_normal().string();
    h.bytes(abs.data(), abs.size());
    if (!hash_file(json_path, h)) return false;
    json_hash[0] = h.a, json_hash[1] = h.b;
    char name[64];
    snprintf(name, sizeof name, "%016llx%016llx.rtscene", (unsigned long long)h.a, (unsigned long long)h.b);
    path_out = (std::filesystem::path(dir) / name).string();
    return true;
}

void put_bytes(std::vector<uint8_t> &v, const void *p, size_t n) { v.insert(v.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
void put_string(std::vector<uint8_t> &v, const std::string &s) {
    const uint32_t n = (uint32_t)s.size();
    put_bytes(v, &n, 4);
    put_bytes(v, s.data(), n);
}

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    bool take(void *out, size_t n) {
        if (!ok || (size_t)(end - p) < n) return ok = false;
        memcpy(out, p, n);
        p += n;
        return true;
    }
    bool string(std::string &s) {
        uint32_t n = 0;
        if (!take(&n, 4) || (size_t)(end - p) < n) return ok = false;
        s.assign((const char *)p, n);
        p += n;
        return true;
    }
};

}  // namespace

struct SceneMapping {
    Mapping file;
    explicit SceneMapping(const std::string &path) : file(path, true) {}
};

bool scene_cache_load(const std::string &json_path, HostScene &out) {
    std::string path;
    uint64_t json_hash[2];
    if (!cache_path_for(json_path, path, json_hash)) return false;
    auto map = std::make_shared<SceneMapping>(path);
    const Mapping &m = map->file;
    if (!m.ok || m.size < sizeof(Header)) return false;
    Header h;
    memcpy(&h, m.data, sizeof h);
    if (h.magic != CACHE_MAGIC || h.version != CACHE_VERSION || h.abi_version != RT_ABI_VERSION || h.n_sections != N_SECTIONS ||
        h.json_hash[0] != json_hash[0] || h.json_hash[1] != json_hash[1] || h.source_stamp != source_stamp())
        return false;
    if (m.size < BODY_START || h.body_bytes != m.size - BODY_START) return false;   // truncated, or longer than what was written
    {
        Hash128 body;
        hash_block(m.data + BODY_START, (size_t)h.body_bytes, body);
        if (body.a != h.body_hash[0] || body.b != h.body_hash[1]) return false;   // damaged or edited
    }
    const void *section[N_SECTIONS];
    uint64_t count[N_SECTIONS];
    for (int k = 0; k < N_SECTIONS; ++k) {
        if (h.struct_bytes[k] != RECORD_BYTES[k] || h.section_offset[k] > m.size || h.section_bytes[k] > m.size - h.section_offset[k] ||
            h.section_bytes[k] % RECORD_BYTES[k] != 0 || h.section_offset[k] % 64 != 0)
            return false;
        section[k] = m.data + h.section_offset[k];
        count[k] = h.section_bytes[k] / RECORD_BYTES[k];
    }
    if (h.tail_offset > m.size || h.tail_bytes > m.size - h.tail_offset) return false;
    Reader r{m.data + h.tail_offset, m.data + h.tail_offset + h.tail_bytes};
    // every file the scene was made from must still have the content it had
    uint32_t n_deps = 0;
    r.take(&n_deps, 4);
    for (uint32_t i = 0; r.ok && i < n_deps; ++i) {
        std::string dep;
        uint64_t want[2];
        if (!r.string(dep) || !r.take(want, 16)) return false;
        Hash128 got;
        if (!hash_file(dep, got) || got.a != want[0] || got.b != want[1]) return false;
    }
    uint32_t n_cams = 0, n_warn = 0;
    r.take(&n_cams, 4);
    std::vector<HostCamera> cameras;
    for (uint32_t i = 0; r.ok && i < n_cams; ++i) {
        HostCamera c;
        uint32_t n_tm = 0;
        r.take(&c.desc, sizeof c.desc);
        r.string(c.image_name);
        r.take(&n_tm, 4);
        for (uint32_t t = 0; r.ok && t < n_tm; ++t) {
            Tonemap tm;
            r.take(&tm.type, 4);
            r.take(&tm.opt1, 4);
            r.take(&tm.opt2, 4);
            r.take(&tm.saturation, 4);
            r.take(&tm.gamma, 4);
            r.string(tm.extension);
            c.tonemaps.push_back(tm);
        }
        cameras.push_back(std::move(c));
    }
    std::vector<std::string> warnings;
    r.take(&n_warn, 4);
    for (uint32_t i = 0; r.ok && i < n_warn; ++i) {
        std::string w;
        r.string(w);
        warnings.push_back(std::move(w));
    }
    if (!r.ok) return false;
    for (int k = 0; k < N_SECTIONS; ++k)   // counts travel as 32-bit integers in the description (texels: 64-bit)
        if (k != 8 && k != 9 && count[k] > 0x7fffffffull) return false;

    out = HostScene();
    out.cameras = std::move(cameras);
    out.warnings = std::move(warnings);
    out.intersection_epsilon = h.intersection_epsilon;
    out.shadow_ray_epsilon = h.shadow_ray_epsilon;
    out.max_depth = h.max_depth;
    memcpy(out.ambient_light, h.ambient_light, sizeof h.ambient_light);
    memcpy(out.background_color, h.background_color, sizeof h.background_color);
    out.background_texture = h.background_texture;
    out.is_probe = h.is_probe;
    rt_scene_desc &d = out.mapped_desc;
    d = rt_scene_desc{};
    d.abi_version = RT_ABI_VERSION;
    d.n_triangles = (int32_t)count[0], d.triangles = (const rt_triangle *)section[0];
    d.n_instances = (int32_t)count[1], d.instances = (const rt_instance *)section[1];
    d.n_spheres = (int32_t)count[2], d.spheres = (const rt_sphere *)section[2];
    d.n_planes = (int32_t)count[3], d.planes = (const rt_plane *)section[3];
    d.n_materials = (int32_t)count[4], d.materials = (const rt_material *)section[4];
    d.n_textures = (int32_t)count[5], d.textures = (const rt_texture *)section[5];
    d.n_texture_refs = (int32_t)count[6], d.texture_refs = (const int32_t *)section[6];
    d.n_images = (int32_t)count[7], d.images = (const rt_image *)section[7];
    d.n_texels_u8 = (int64_t)count[8], d.texels_u8 = (const uint8_t *)section[8];
    d.n_texels_f32 = (int64_t)count[9], d.texels_f32 = (const float *)section[9];
    d.n_lights = (int32_t)count[10], d.lights = (const rt_light *)section[10];
    d.n_bvh_nodes = (int32_t)count[11], d.bvh_nodes = (const rt_bvh_node *)section[11];
    d.n_bvh_prims = (int32_t)count[12], d.bvh_prims = (const rt_bvh_prim *)section[12];
    d.intersection_epsilon = h.intersection_epsilon;
    d.shadow_ray_epsilon = h.shadow_ray_epsilon;
    d.max_depth = h.max_depth;
    memcpy(d.ambient_light, h.ambient_light, sizeof h.ambient_light);
    memcpy(d.background_color, h.background_color, sizeof h.background_color);
    d.background_texture = h.background_texture;
    d.is_probe = h.is_probe;
    out.mapping = map;
    return true;
}

void scene_cache_store(const std::string &json_path, const HostScene &scene) {
    std::string path;
    uint64_t json_hash[2];
    if (scene.mapping || !cache_path_for(json_path, path, json_hash)) return;
    std::error_code ec;
    std::filesystem::create_directories(cache_dir(), ec);
    const rt_scene_desc d = scene.desc();
    const void *section[N_SECTIONS] = {d.triangles, d.instances, d.spheres,    d.planes, d.materials, d.textures, d.texture_refs,
                                       d.images,    d.texels_u8, d.texels_f32, d.lights, d.bvh_nodes, d.bvh_prims};
    const uint64_t count[N_SECTIONS] = {(uint64_t)d.n_triangles, (uint64_t)d.n_instances, (uint64_t)d.n_spheres,    (uint64_t)d.n_planes, (uint64_t)d.n_materials,
                                        (uint64_t)d.n_textures,  (uint64_t)d.n_texture_refs, (uint64_t)d.n_images,  (uint64_t)d.n_texels_u8,
                                        (uint64_t)d.n_texels_f32, (uint64_t)d.n_lights,   (uint64_t)d.n_bvh_nodes, (uint64_t)d.n_bvh_prims};
    Header h{};
    h.magic = CACHE_MAGIC, h.version = CACHE_VERSION, h.abi_version = RT_ABI_VERSION, h.n_sections = N_SECTIONS;
    h.json_hash[0] = json_hash[0], h.json_hash[1] = json_hash[1];
    h.source_stamp = source_stamp();
    uint64_t at = BODY_START;
    for (int k = 0; k < N_SECTIONS; ++k) {
        h.struct_bytes[k] = RECORD_BYTES[k];
        h.section_offset[k] = at;
        h.section_bytes[k] = count[k] * RECORD_BYTES[k];
        at = (at + h.section_bytes[k] + 63) / 64 * 64;
    }
    std::vector<uint8_t> tail;
    const uint32_t n_deps = (uint32_t)scene.dependencies.size();
    put_bytes(tail, &n_deps, 4);
    for (const std::string &dep : scene.dependencies) {
        Hash128 dh;
        if (!hash_file(dep, dh)) return;   // a dependency that cannot be read again: no cache for this scene
        put_string(tail, dep);
        const uint64_t v[2] = {dh.a, dh.b};
        put_bytes(tail, v, 16);
    }
    const uint32_t n_cams = (uint32_t)scene.cameras.size();
    put_bytes(tail, &n_cams, 4);
    for (const HostCamera &c : scene.cameras) {
        put_bytes(tail, &c.desc, sizeof c.desc);
        put_string(tail, c.image_name);
        const uint32_t n_tm = (uint32_t)c.tonemaps.size();
        put_bytes(tail, &n_tm, 4);
        for (const Tonemap &tm : c.tonemaps) {
            put_bytes(tail, &tm.type, 4);
            put_bytes(tail, &tm.opt1, 4);
            put_bytes(tail, &tm.opt2, 4);
            put_bytes(tail, &tm.saturation, 4);
            put_bytes(tail, &tm.gamma, 4);
            put_string(tail, tm.extension);
        }
    }
    const uint32_t n_warn = (uint32_t)scene.warnings.size();
    put_bytes(tail, &n_warn, 4);
    for (const std::string &w : scene.warnings) put_string(tail, w);
    h.tail_offset = at, h.tail_bytes = tail.size();
    h.intersection_epsilon = scene.intersection_epsilon, h.shadow_ray_epsilon = scene.shadow_ray_epsilon, h.max_depth = scene.max_depth;
    memcpy(h.ambient_light, scene.ambient_light, sizeof h.ambient_light);
    memcpy(h.background_color, scene.background_color, sizeof h.background_color);
    h.background_texture = scene.background_texture, h.is_probe = scene.is_probe;

    // The body as it will lie in the file (sections at their offsets, zero padding between, the tail last); hashed, then
    // written under a private name and renamed into place: a reader never sees half a file.
    h.body_bytes = h.tail_offset + h.tail_bytes - BODY_START;
    std::vector<uint8_t> body;
    try {
        body.assign((size_t)h.body_bytes, 0);
    } catch (const std::bad_alloc &) {
        return;   // no memory for a second copy of the scene: no cache for it
    }
    for (int k = 0; k < N_SECTIONS; ++k)
        if (h.section_bytes[k]) memcpy(body.data() + (h.section_offset[k] - BODY_START), section[k], (size_t)h.section_bytes[k]);
    if (!tail.empty()) memcpy(body.data() + (h.tail_offset - BODY_START), tail.data(), tail.size());
    {
        Hash128 bh;
        hash_block(body.data(), body.size(), bh);
        h.body_hash[0] = bh.a, h.body_hash[1] = bh.b;
    }
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1;
    static const uint8_t zeros[64] = {};
    if (ok && BODY_START > sizeof h) ok = fwrite(zeros, 1, BODY_START - sizeof h, f) == BODY_START - sizeof h;
    if (ok && !body.empty()) ok = fwrite(body.data(), 1, body.size(), f) == body.size();
    ok = fclose(f) == 0 && ok;
    if (ok)
        std::filesystem::rename(tmp, path, ec);
    if (!ok || ec) std::filesystem::remove(tmp, ec);
}

}  // namespace rth
