// PLY mesh reader (reference src/ply_loader.cpp:63-433): ascii and binary_little_endian, arbitrary
// scalar vertex properties (x y z nx ny nz, u|s, v|t are used, the rest skipped), polygon faces
// triangulated as fans, area-weighted vertex normals when smooth shading is requested and the file
// has none.  A file that carries normals always yields smooth triangles (src/ply_loader.cpp:406-418).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "hmath.h"
#include "scene.h"

namespace rth {

namespace {

std::string trimmed(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), ::tolower);
    return s;
}

struct Cursor {
    const unsigned char *p, *end;
    bool take(void *dst, size_t n) {
        if ((size_t)(end - p) < n) return false;
        memcpy(dst, p, n);
        p += n;
        return true;
    }
};

// Scalar types and vertex property names are resolved once, when the header is read (the reference compares
// strings for every value, src/ply_loader.cpp:185-294).
enum Scalar { S_UNKNOWN, S_F32, S_F64, S_U8, S_I8, S_U16, S_I16, S_I32, S_U32, S_LIST };
enum Slot { P_NONE, P_X, P_Y, P_Z, P_NX, P_NY, P_NZ, P_U, P_V };

Scalar scalar_of(const std::string &t) {
    if (t == "list") return S_LIST;
    if (t == "float" || t == "float32") return S_F32;
    if (t == "double" || t == "float64") return S_F64;
    if (t == "uchar" || t == "uint8") return S_U8;
    if (t == "char" || t == "int8") return S_I8;
    if (t == "ushort" || t == "uint16") return S_U16;
    if (t == "short" || t == "int16") return S_I16;
    if (t == "int" || t == "int32" || t == "int32_t") return S_I32;
    if (t == "uint" || t == "uint32" || t == "uint32_t") return S_U32;
    return S_UNKNOWN;
}

Slot slot_of(const std::string &name) {
    if (name == "x") return P_X;
    if (name == "y") return P_Y;
    if (name == "z") return P_Z;
    if (name == "nx") return P_NX;
    if (name == "ny") return P_NY;
    if (name == "nz") return P_NZ;
    if (name == "u" || name == "s") return P_U;
    if (name == "v" || name == "t") return P_V;
    return P_NONE;
}

bool read_scalar(Cursor &c, Scalar t, double &v) {
    switch (t) {
        case S_F32: { float x; if (!c.take(&x, 4)) return false; v = x; return true; }
        case S_F64: { double x; if (!c.take(&x, 8)) return false; v = x; return true; }
        case S_U8: { uint8_t x; if (!c.take(&x, 1)) return false; v = x; return true; }
        case S_I8: { int8_t x; if (!c.take(&x, 1)) return false; v = x; return true; }
        case S_U16: { uint16_t x; if (!c.take(&x, 2)) return false; v = x; return true; }
        case S_I16: { int16_t x; if (!c.take(&x, 2)) return false; v = x; return true; }
        case S_I32: { int32_t x; if (!c.take(&x, 4)) return false; v = x; return true; }
        case S_U32: { uint32_t x; if (!c.take(&x, 4)) return false; v = x; return true; }
        default: return false;
    }
}

}  // namespace

bool load_ply(const std::string &path, std::vector<float> &vertices, HostScene &scene, bool smooth_shading, PlyResult &res,
              std::string &err) {
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) {
        err = "Error: could not open PLY file " + path;
        return false;
    }
    std::string data((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    auto getline = [&](std::string &line) -> bool {
        if (pos >= data.size()) return false;
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        line.assign(data, pos, e - pos);
        pos = e + 1;
        return true;
    };

    size_t vertex_count = 0, face_count = 0;
    std::string index_type = "int", count_type = "uchar", format = "ascii", current;
    std::vector<std::pair<std::string, std::string>> vprops;   // (type, name)
    bool has_normals = false, has_uvs = false;
    std::string line;
    while (getline(line)) {
        line = trimmed(line);
        if (line.empty()) continue;
        std::istringstream iss(line);
        std::string token;
        if (!(iss >> token)) continue;
        if (token == "format") {
            iss >> format;
        } else if (token == "element") {
            iss >> current;
            if (current == "vertex")
                iss >> vertex_count;
            else if (current == "face")
                iss >> face_count;
        } else if (token == "property") {
            if (current == "vertex") {
                std::string type, name;
                iss >> type >> name;
                if (type == "list") {
                    std::string it;
                    iss >> it >> name;
                    vprops.emplace_back("list", name);
                } else {
                    vprops.emplace_back(type, name);
                }
                if (name == "nx" || name == "ny" || name == "nz") has_normals = true;
                if (name == "u" || name == "v" || name == "s" || name == "t") has_uvs = true;
            } else if (current == "face") {
                std::string next;
                iss >> next;
                if (next == "list") {
                    std::string ct, it, name;
                    iss >> ct >> it >> name;
                    if (name == "vertex_indices" || name == "vertex_index") {
                        count_type = ct;
                        index_type = it;
                    }
                }
            }
        } else if (token == "end_header") {
            break;
        }
    }

    // per vertex property: (binary scalar type of the lower-cased type name, destination slot)
    std::vector<std::pair<Scalar, Slot>> vcodes;
    for (const auto &prop : vprops) vcodes.emplace_back(prop.first == "list" ? S_LIST : scalar_of(lower(prop.first)), slot_of(prop.second));
    const Scalar count_code = scalar_of(count_type), index_code = scalar_of(index_type);

    const size_t start_index = vertices.size() / 3;
    std::vector<V3> normals;
    std::vector<std::pair<float, float>> uvs;
    const bool keep_normals = has_normals || smooth_shading;
    if (keep_normals) normals.reserve(vertex_count);
    if (has_uvs) uvs.reserve(vertex_count);
    vertices.reserve(vertices.size() + vertex_count * 3);
    const bool is_ascii = format.find("ascii") != std::string::npos;
    Cursor cur{(const unsigned char *)data.data() + std::min(pos, data.size()), (const unsigned char *)data.data() + data.size()};

    for (size_t vi = 0; vi < vertex_count; ++vi) {
        float vx = 0, vy = 0, vz = 0, nx = 0, ny = 0, nz = 0, vu = 0, vv = 0;
        auto assign = [&](Slot slot, float val) {
            switch (slot) {
                case P_X: vx = val; break;
                case P_Y: vy = val; break;
                case P_Z: vz = val; break;
                case P_NX: nx = val; break;
                case P_NY: ny = val; break;
                case P_NZ: nz = val; break;
                case P_U: vu = val; break;
                case P_V: vv = val; break;
                default: break;
            }
        };
        if (is_ascii) {
            if (!getline(line)) {
                err = "Unexpected EOF in " + path;
                return false;
            }
            std::istringstream viss(line);
            for (const auto &code : vcodes) {
                std::string tok;
                if (!(viss >> tok)) break;
                if (code.first == S_LIST) continue;
                assign(code.second, strtof(tok.c_str(), nullptr));
            }
        } else {
            for (const auto &code : vcodes) {
                if (code.first == S_LIST) {   // the reference skips a uchar count followed by uchar items
                    uint8_t n = 0;
                    if (!cur.take(&n, 1)) break;
                    cur.p = std::min(cur.p + n, cur.end);
                    continue;
                }
                double v;
                if (read_scalar(cur, code.first, v)) {
                    assign(code.second, (float)v);
                } else {
                    cur.p = std::min(cur.p + 4, cur.end);   // unknown type: the reference seeks 4 bytes
                }
            }
        }
        vertices.push_back(vx);
        vertices.push_back(vy);
        vertices.push_back(vz);
        if (keep_normals) normals.emplace_back(nx, ny, nz);
        if (has_uvs) uvs.emplace_back(vu, vv);
    }
    if (!is_ascii) pos = (size_t)(cur.p - (const unsigned char *)data.data());

    std::vector<int32_t> tris;   // 3 indices per triangle, local to this file
    tris.reserve(face_count * 3);
    std::vector<int32_t> face;
    for (size_t fi = 0; fi < face_count; ++fi) {
        uint32_t n = 0;
        face.clear();
        if (is_ascii) {
            if (!getline(line)) break;
            std::istringstream fiss(line);
            fiss >> n;
            int32_t idx;
            for (uint32_t j = 0; j < n; ++j)
                if (fiss >> idx) face.push_back(idx);
        } else {
            double c;
            if (!read_scalar(cur, count_code, c)) break;
            n = (uint32_t)c;
            bool good = true;
            for (uint32_t j = 0; j < n; ++j) {
                double v;
                if (!read_scalar(cur, index_code, v)) {
                    good = false;
                    break;
                }
                face.push_back((int32_t)v);
            }
            if (!good) break;
        }
        if (n < 3 || face.size() < n) continue;
        for (size_t i = 1; i + 1 < n; ++i) {
            tris.push_back(face[0]);
            tris.push_back(face[i]);
            tris.push_back(face[i + 1]);
        }
    }
    for (int32_t idx : tris)
        if (idx < 0 || (size_t)idx >= vertex_count) {
            err = "PLY face index out of range in " + path;
            return false;
        }

    auto vert = [&](int32_t i) {
        size_t k = (start_index + (size_t)i) * 3;
        return V3(vertices[k], vertices[k + 1], vertices[k + 2]);
    };
    const size_t ntri = tris.size() / 3;
    if (smooth_shading && !has_normals) {
        normals.assign(vertex_count, V3(0, 0, 0));
        for (size_t k = 0; k < ntri; ++k) {
            V3 v0 = vert(tris[k * 3]), v1 = vert(tris[k * 3 + 1]), v2 = vert(tris[k * 3 + 2]);
            V3 n = cross(v1 - v0, v2 - v0);
            for (int j = 0; j < 3; ++j) normals[tris[k * 3 + j]] = normals[tris[k * 3 + j]] + n;
        }
        for (auto &n : normals) n = normalize(n);
        has_normals = true;
    }
    scene.triangles.reserve(scene.triangles.size() + ntri);
    for (size_t k = 0; k < ntri; ++k) {
        rt_triangle t{};
        V3 v[3] = {vert(tris[k * 3]), vert(tris[k * 3 + 1]), vert(tris[k * 3 + 2])};
        memcpy(t.c1, &v[0], 12);
        memcpy(t.c2, &v[1], 12);
        memcpy(t.c3, &v[2], 12);
        if (has_uvs)
            for (int j = 0; j < 3; ++j) {
                t.tex_u[j] = uvs[tris[k * 3 + j]].first;
                t.tex_v[j] = uvs[tris[k * 3 + j]].second;
            }
        if (has_normals) {
            t.smooth = 1;
            memcpy(t.n1, &normals[tris[k * 3]], 12);
            memcpy(t.n2, &normals[tris[k * 3 + 1]], 12);
            memcpy(t.n3, &normals[tris[k * 3 + 2]], 12);
        }
        scene.triangles.push_back(t);
    }
    res.vertex_count = vertex_count;
    res.tri_count = ntri;
    res.has_normals = has_normals;
    res.has_uvs = has_uvs;
    return true;
}

}  // namespace rth
