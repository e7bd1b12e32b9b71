// Image files on the host side of the boundary: texture decoding (PNG via zlib, OpenEXR scan-line
// files with NONE/ZIPS/ZIP compression) and output encoding (8-bit RGB PNG, 32-bit float RGB EXR).
// The reference delegates these to stb_image / stb_image_write / tinyexr (reference
// src/texture.cpp:18-54, src/main.cpp:561-607); they are outside the rendering hot path.
#include "image_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "scene.h"

namespace rth {

namespace {

bool read_all(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) {
        fclose(f);
        return false;
    }
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool zlib_inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &dst, size_t expected) {
    dst.resize(expected);
    uLongf len = (uLongf)expected;
    int rc = uncompress(dst.data(), &len, src, (uLong)n);
    if (rc != Z_OK) return false;
    dst.resize(len);
    return true;
}

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

// PNG -> 8-bit RGB.  Colour types 0,2,3,4,6; bit depths 1..16; plain or Adam7-interlaced.
bool decode_png(const std::vector<uint8_t> &file, int &w, int &h, std::vector<uint8_t> &rgb, std::string &err) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) {
        err = "not a PNG";
        return false;
    }
    size_t p = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    w = h = 0;
    while (p + 12 <= file.size()) {
        uint32_t len = be32(&file[p]);
        const uint8_t *type = &file[p + 4];
        const uint8_t *data = &file[p + 8];
        if (p + 12 + len > file.size()) break;
        if (!memcmp(type, "IHDR", 4)) {
            w = (int)be32(data);
            h = (int)be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        p += 12 + len;
    }
    if (w <= 0 || h <= 0 || (uint64_t)w * (uint64_t)h > (1ull << 28) || interlace > 1) {
        err = "unsupported PNG (bad size or interlace method)";
        return false;
    }
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16)) {
        err = "bad PNG colour type or bit depth";
        return false;
    }
    const int bpp_bits = channels * depth;
    const int bpp = std::max(1, bpp_bits / 8);
    // the image is one pass, or the seven Adam7 passes (x origin, y origin, x spacing, y spacing), each filtered on its own
    struct Pass {
        int x0, y0, dx, dy;
    };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass *passes = interlace ? adam7 : &whole;
    const int n_passes = interlace ? 7 : 1;
    auto pass_size = [&](const Pass &ps, int &pw, int &ph, size_t &stride) {
        pw = (w - ps.x0 + ps.dx - 1) / ps.dx;
        ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        stride = ((size_t)pw * bpp_bits + 7) / 8;
    };
    size_t total = 0;
    for (int i = 0; i < n_passes; ++i) {
        int pw, ph;
        size_t stride;
        pass_size(passes[i], pw, ph, stride);
        if (pw > 0 && ph > 0) total += (stride + 1) * (size_t)ph;
    }
    std::vector<uint8_t> raw;
    if (!zlib_inflate(idat.data(), idat.size(), raw, total) || raw.size() != total) {
        err = "PNG inflate failed";
        return false;
    }
    rgb.resize((size_t)w * h * 3);
    size_t at = 0;
    std::vector<uint8_t> img;
    for (int i = 0; i < n_passes; ++i) {
        const Pass &ps = passes[i];
        int pw, ph;
        size_t stride;
        pass_size(ps, pw, ph, stride);
        if (pw <= 0 || ph <= 0) continue;
        img.assign(stride * (size_t)ph, 0);
        for (int y = 0; y < ph; ++y) {
            const uint8_t *src = &raw[at + (stride + 1) * y];
            uint8_t *dst = &img[stride * y];
            const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
            int filter = src[0];
            for (size_t x = 0; x < stride; ++x) {
                int a = x >= (size_t)bpp ? dst[x - bpp] : 0;
                int b = up ? up[x] : 0;
                int c = (up && x >= (size_t)bpp) ? up[x - bpp] : 0;
                int v = src[x + 1];
                switch (filter) {
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) / 2; break;
                    case 4: v += paeth(a, b, c); break;
                    default: break;
                }
                dst[x] = (uint8_t)v;
            }
        }
        at += (stride + 1) * (size_t)ph;
        for (int y = 0; y < ph; ++y) {
            const uint8_t *row = &img[stride * y];
            for (int x = 0; x < pw; ++x) {
                int s[4] = {0, 0, 0, 255};
                for (int c = 0; c < channels; ++c) {
                    if (depth == 8) {
                        s[c] = row[x * channels + c];
                    } else if (depth == 16) {
                        s[c] = row[(x * channels + c) * 2];
                    } else {
                        int bit = (x * channels + c) * depth;
                        int v = (row[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
                        s[c] = ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
                    }
                }
                uint8_t *o = &rgb[((size_t)(y * ps.dy + ps.y0) * w + (size_t)(x * ps.dx + ps.x0)) * 3];
                if (ctype == 3) {
                    size_t k = (size_t)s[0] * 3;
                    o[0] = k + 2 < palette.size() ? palette[k] : 0;
                    o[1] = k + 2 < palette.size() ? palette[k + 1] : 0;
                    o[2] = k + 2 < palette.size() ? palette[k + 2] : 0;
                } else if (channels <= 2) {
                    o[0] = o[1] = o[2] = (uint8_t)s[0];
                } else {
                    o[0] = (uint8_t)s[0];
                    o[1] = (uint8_t)s[1];
                    o[2] = (uint8_t)s[2];
                }
            }
        }
    }
    return true;
}

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h >> 15) << 31, exp = (h >> 10) & 31, man = h & 1023, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do {
                ++e;
                man <<= 1;
            } while (!(man & 1024));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 1023) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

// OpenEXR scan-line image -> float RGBA, top row first (as tinyexr's LoadEXR returns it).
bool decode_exr(const std::vector<uint8_t> &file, int &w, int &h, std::vector<float> &rgba, std::string &err) {
    if (file.size() < 8 || file[0] != 0x76 || file[1] != 0x2f || file[2] != 0x31 || file[3] != 0x01) {
        err = "not an EXR";
        return false;
    }
    if (file[5] & 0x02) {
        err = "tiled EXR not supported";
        return false;
    }
    // Every size and offset below comes from the file: each is checked against what the file holds before it is used.
    size_t p = 8;
    struct Chan {
        std::string name;
        int type;
    };
    std::vector<Chan> chans;
    int compression = 0;
    int64_t xmin = 0, ymin = 0, xmax = -1, ymax = -1;
    auto cstr = [&](std::string &s) {   // false when the string runs off the end of the file
        s.clear();
        while (p < file.size() && file[p]) s += (char)file[p++];
        if (p >= file.size()) return false;
        ++p;
        return true;
    };
    bool header_ended = false;
    while (p < file.size()) {
        if (!file[p]) {
            header_ended = true;
            break;
        }
        std::string name, type;
        if (!cstr(name) || !cstr(type) || p + 4 > file.size()) break;
        uint32_t size;
        memcpy(&size, &file[p], 4);
        p += 4;
        if ((size_t)size > file.size() - p) break;   // the attribute's value must lie inside the file
        const uint8_t *v = &file[p];
        if (name == "channels") {
            size_t q = 0;
            while (q < size && v[q]) {
                Chan c;
                while (q < size && v[q]) c.name += (char)v[q++];
                ++q;   // the name's terminator
                if (q + 16 > size) {
                    err = "bad EXR channel list";
                    return false;
                }
                int32_t t;
                memcpy(&t, v + q, 4);
                if (t < 0 || t > 2) {
                    err = "bad EXR channel type";
                    return false;
                }
                c.type = t;
                q += 16;
                chans.push_back(c);
            }
        } else if (name == "compression") {
            if (size < 1) break;
            compression = v[0];
        } else if (name == "dataWindow") {
            if (size < 16) break;
            int32_t b[4];
            memcpy(b, v, 16);
            xmin = b[0];
            ymin = b[1];
            xmax = b[2];
            ymax = b[3];
        }
        p += size;
    }
    if (!header_ended) {
        err = "truncated EXR header";
        return false;
    }
    ++p;
    const int64_t w64 = xmax - xmin + 1, h64 = ymax - ymin + 1;
    if (w64 <= 0 || h64 <= 0 || chans.empty() || w64 > (1 << 16) || h64 > (1 << 16) || w64 * h64 > (int64_t)1 << 28) {   // the bound decode_png uses
        err = "bad EXR header";
        return false;
    }
    w = (int)w64;
    h = (int)h64;
    int lines_per_block = compression == 0 || compression == 2 ? 1 : compression == 3 ? 16 : 0;
    if (!lines_per_block) {
        err = "EXR compression not supported (only NONE, ZIPS, ZIP)";
        return false;
    }
    size_t bytes_per_line = 0;
    for (auto &c : chans) bytes_per_line += (size_t)w * (c.type == 1 ? 2 : 4);
    int nblocks = (h + lines_per_block - 1) / lines_per_block;
    if (p > file.size() || (size_t)nblocks * 8 > file.size() - p) {
        err = "truncated EXR";
        return false;
    }
    rgba.assign((size_t)w * h * 4, 0.0f);
    for (size_t i = 3; i < rgba.size(); i += 4) rgba[i] = 1.0f;
    std::vector<uint8_t> tmp, block;
    for (int b = 0; b < nblocks; ++b) {
        uint64_t off;
        memcpy(&off, &file[p + (size_t)b * 8], 8);
        if (off > file.size() || file.size() - off < 8) {
            err = "truncated EXR";
            return false;
        }
        int32_t y0, dsize;
        memcpy(&y0, &file[off], 4);
        memcpy(&dsize, &file[off + 4], 4);
        if (dsize < 0 || (uint64_t)dsize > file.size() - off - 8) {
            err = "truncated EXR";
            return false;
        }
        if (y0 < ymin || y0 > ymax) {
            err = "EXR block outside the data window";
            return false;
        }
        int lines = (int)std::min<int64_t>(lines_per_block, ymax - y0 + 1);
        size_t raw_size = bytes_per_line * (size_t)lines;
        if (!(compression != 0 && (size_t)dsize < raw_size) && (size_t)dsize != raw_size) {   // stored as it is: the sizes must agree
            err = "EXR block size does not match its scan lines";
            return false;
        }
        const uint8_t *src = &file[off + 8];
        if (compression != 0 && (size_t)dsize < raw_size) {
            if (!zlib_inflate(src, (size_t)dsize, tmp, raw_size) || tmp.size() != raw_size) {
                err = "EXR inflate failed";
                return false;
            }
            for (size_t i = 1; i < raw_size; ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);   // predictor
            block.resize(raw_size);
            size_t half = (raw_size + 1) / 2;
            for (size_t i = 0; i < raw_size; ++i) block[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];   // de-interleave
            src = block.data();
        }
        for (int l = 0; l < lines; ++l) {
            const int y = (int)(y0 - ymin) + l;
            const uint8_t *q = src + bytes_per_line * (size_t)l;
            for (auto &c : chans) {
                int slot = c.name == "R" ? 0 : c.name == "G" ? 1 : c.name == "B" ? 2 : c.name == "A" ? 3 : (chans.size() == 1 ? -2 : -1);
                for (int x = 0; x < w; ++x) {
                    float v;
                    if (c.type == 1) {
                        uint16_t hv;
                        memcpy(&hv, q + (size_t)x * 2, 2);
                        v = half_to_float(hv);
                    } else if (c.type == 2) {
                        memcpy(&v, q + (size_t)x * 4, 4);
                    } else {
                        uint32_t u;
                        memcpy(&u, q + (size_t)x * 4, 4);
                        v = (float)u;
                    }
                    float *o = &rgba[((size_t)y * w + x) * 4];
                    if (slot >= 0)
                        o[slot] = v;
                    else if (slot == -2)
                        o[0] = o[1] = o[2] = v;
                }
                q += (size_t)w * (c.type == 1 ? 2 : 4);
            }
        }
    }
    return true;
}

uint32_t crc_of(const uint8_t *p, size_t n, uint32_t c = 0) { return (uint32_t)crc32(c, p, (uInt)n); }

void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}

void png_chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data) {
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc_of(&out[start], out.size() - start));
}

}  // namespace

bool decode_image_rgb8(const std::vector<uint8_t> &file, int &w, int &h, std::vector<uint8_t> &rgb, std::string &err) {
    return decode_png(file, w, h, rgb, err) || decode_jpeg(file, w, h, rgb, err);
}

bool load_image_file(const std::string &path, HostScene &scene, rt_image &out, std::string &err) {
    std::vector<uint8_t> file;
    if (!read_all(path, file)) {
        err = "cannot read file";
        return false;
    }
    int w = 0, h = 0;
    // the reference decides by extension: ".exr"/".EXR" -> float RGBA, otherwise 8-bit RGB (src/texture.cpp:9-16)
    bool is_exr = path.size() >= 4 && (path.substr(path.size() - 4) == ".exr" || path.substr(path.size() - 4) == ".EXR");
    if (is_exr) {
        std::vector<float> rgba;
        if (!decode_exr(file, w, h, rgba, err)) return false;
        out.width = w;
        out.height = h;
        out.is_hdr = 1;
        out.offset = (int32_t)scene.texels_f32.size();
        scene.texels_f32.insert(scene.texels_f32.end(), rgba.begin(), rgba.end());
        return true;
    }
    std::vector<uint8_t> rgb;
    if (!decode_image_rgb8(file, w, h, rgb, err)) return false;
    out.width = w;
    out.height = h;
    out.is_hdr = 0;
    out.offset = (int32_t)scene.texels_u8.size();
    scene.texels_u8.insert(scene.texels_u8.end(), rgb.begin(), rgb.end());
    return true;
}

bool write_png_rgb8(const std::string &path, int w, int h, const uint8_t *rgb, bool flip_vertically) {
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * 3 + 1));
    for (int y = 0; y < h; ++y) {
        int sy = flip_vertically ? h - 1 - y : y;
        uint8_t *dst = &raw[(size_t)y * ((size_t)w * 3 + 1)];
        dst[0] = 0;
        memcpy(dst + 1, rgb + (size_t)sy * w * 3, (size_t)w * 3);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    // zlib level 3 unless RT_PNG_LEVEL (1 .. 9) says otherwise: the pixels are the same at every level, size and time are not
    // (a 1080p frame of the 70 K-triangle scene: 0.07 MB in 16 ms at level 6, 0.13 MB in 9 ms at 3; the reference's writer,
    // stb_image_write, makes 0.18 MB of it)
    static const int level = [] {
        const char *e = getenv("RT_PNG_LEVEL");
        const int v = e ? atoi(e) : 3;
        return v >= 1 && v <= 9 ? v : 3;
    }();
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), level) != Z_OK) return false;
    comp.resize(clen);
    std::vector<uint8_t> out = {137, 80, 78, 71, 13, 10, 26, 10};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w);
    put_be32(ihdr, (uint32_t)h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", comp);
    png_chunk(out, "IEND", {});
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

// 3-channel 32-bit float scan-line EXR, no compression.  Rows are written in buffer order (the
// reference hands its bottom-row-first buffer to SaveEXR unflipped, src/main.cpp:565).
bool write_exr_rgb32f(const std::string &path, int w, int h, const float *rgb) {
    std::vector<uint8_t> out = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    auto put_str = [&](const char *s) { out.insert(out.end(), s, s + strlen(s) + 1); };
    auto put_i32 = [&](int32_t v) {
        uint8_t b[4];
        memcpy(b, &v, 4);
        out.insert(out.end(), b, b + 4);
    };
    auto put_f32 = [&](float v) {
        uint8_t b[4];
        memcpy(b, &v, 4);
        out.insert(out.end(), b, b + 4);
    };
    put_str("channels");
    put_str("chlist");
    put_i32(3 * 18 + 1);
    for (const char *c : {"B", "G", "R"}) {
        put_str(c);
        put_i32(2);   // FLOAT
        out.insert(out.end(), {0, 0, 0, 0});
        put_i32(1);
        put_i32(1);
    }
    out.push_back(0);
    put_str("compression");
    put_str("compression");
    put_i32(1);
    out.push_back(0);
    for (const char *name : {"dataWindow", "displayWindow"}) {
        put_str(name);
        put_str("box2i");
        put_i32(16);
        put_i32(0);
        put_i32(0);
        put_i32(w - 1);
        put_i32(h - 1);
    }
    put_str("lineOrder");
    put_str("lineOrder");
    put_i32(1);
    out.push_back(0);
    put_str("pixelAspectRatio");
    put_str("float");
    put_i32(4);
    put_f32(1.0f);
    put_str("screenWindowCenter");
    put_str("v2f");
    put_i32(8);
    put_f32(0.0f);
    put_f32(0.0f);
    put_str("screenWindowWidth");
    put_str("float");
    put_i32(4);
    put_f32(1.0f);
    out.push_back(0);
    size_t table = out.size();
    size_t line_bytes = 8 + (size_t)w * 12;
    out.resize(table + (size_t)h * 8 + (size_t)h * line_bytes);
    for (int y = 0; y < h; ++y) {
        uint64_t off = table + (size_t)h * 8 + (size_t)y * line_bytes;
        memcpy(&out[table + (size_t)y * 8], &off, 8);
        int32_t yy = y, sz = w * 12;
        memcpy(&out[off], &yy, 4);
        memcpy(&out[off + 4], &sz, 4);
        float *dst = (float *)&out[off + 8];
        for (int c = 0; c < 3; ++c)   // channel order B, G, R
            for (int x = 0; x < w; ++x) {
                float v = rgb[((size_t)y * w + x) * 3 + (2 - c)];
                memcpy(&dst[(size_t)c * w + x], &v, 4);
            }
    }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

bool read_exr_rgba(const std::string &path, int &w, int &h, std::vector<float> &rgba, std::string &err) {
    std::vector<uint8_t> file;
    if (!read_all(path, file)) {
        err = "cannot read file";
        return false;
    }
    return decode_exr(file, w, h, rgba, err);
}

}  // namespace rth
