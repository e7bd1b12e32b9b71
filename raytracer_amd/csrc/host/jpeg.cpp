// JPEG texture decoding: baseline and progressive Huffman JPEG (ITU T.81), 8-bit, 1 / 3 / 4 components, any
// sampling factors, restart intervals.
//
// The reference loads 8-bit textures with stbi_load(path, ..., 3) (reference src/texture.cpp:46; stb_image v2.30 under
// third_party/stb).  A texture's texels enter shading directly, so the decoded bytes have to be the ones that decoder
// produces; the steps that decide them are restated here:
//   * dequantised coefficients are kept as 16-bit values,
//   * the inverse DCT is the 12-bit fixed-point "islow" variant (columns rounded to 2 extra bits with +512 >> 10 and
//     saturated to 16 bits, rows with +65536 + (128 << 17) >> 17, then clamped to 0..255),
//   * chroma is upsampled with the centred triangle filter for 2x1, 1x2 and 2x2 ((3a + b + 2) >> 2,
//     (3t0 + t1 + 8) >> 4), by sample repetition for any other ratio,
//   * YCbCr -> RGB in 20-bit fixed point with constants rounded to 12 bits and the Cb term of green truncated to its
//     upper 16 bits; Adobe APP14 transforms 0 (RGB / CMYK) and 2 (YCCK) are honoured.
// tests/test_host.py compares the output byte for byte with fixtures decoded by the reference's own decoder.
#include <algorithm>
#include <cstdint>
#include <cstring>

#include "image_io.h"

namespace rth {

namespace {

const uint8_t ZIGZAG[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                 39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffmanTable {
    bool present = false;
    uint8_t lengths[256];    // code length of symbol number i (symbols in order of increasing code)
    uint8_t symbols[256];
    int n_symbols = 0;
    int first_code[18];      // first code of each length
    int first_index[18];     // index of its symbol
    int count[18];

    bool build(const int *counts, const uint8_t *values, int total) {
        n_symbols = total;
        if (total > 256) return false;
        memcpy(symbols, values, (size_t)total);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            first_code[len] = code;
            first_index[len] = k;
            count[len] = counts[len - 1];
            for (int i = 0; i < counts[len - 1]; ++i) lengths[k++] = (uint8_t)len;
            code += counts[len - 1];
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int dc_pred = 0;
    int x = 0, y = 0;          // size in samples
    int w2 = 0, h2 = 0;        // size of the stored plane (whole MCUs)
    int blocks_w = 0, blocks_h = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;   // progressive: 64 per block, natural order
};

class Decoder {
public:
    Decoder(const std::vector<uint8_t> &f) : file(f) {}

    bool decode(int &w, int &h, std::vector<uint8_t> &rgb, std::string &err) {
        if (!run()) {
            err = error.empty() ? "corrupt JPEG" : error;
            return false;
        }
        w = width;
        h = height;
        convert(rgb);
        return true;
    }

private:
    const std::vector<uint8_t> &file;
    size_t pos = 0;
    std::string error;
    int width = 0, height = 0, n_comp = 0;
    bool progressive = false, jfif = false, rgb_ids = false;
    int adobe_transform = -1;
    int h_max = 1, v_max = 1, mcu_w = 0, mcu_h = 0, mcus_x = 0, mcus_y = 0;
    int restart_interval = 0;
    uint16_t quant[4][64];   // natural order
    HuffmanTable dc_tables[4], ac_tables[4];
    Component comp[4];
    // entropy-coded segment state
    uint32_t bit_buffer = 0;
    int bit_count = 0;
    int pending_marker = -1;
    bool no_more = false;
    int scan_n = 0, scan_order[4];
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;

    bool fail(const char *what) {
        if (error.empty()) error = what;
        return false;
    }
    int byte() { return pos < file.size() ? file[pos++] : 0; }
    int word() {
        int a = byte();
        return (a << 8) | byte();
    }

    // ---- bits ------------------------------------------------------------------------------------------------
    // The entropy-coded bytes are consumed exactly as the reference's decoder consumes them (third_party/stb/
    // stb_image.h:2072-2190), which fixes what a damaged or truncated scan decodes to: the buffer is topped up to more than
    // 24 bits; the top-up that runs into a marker stops short, later ones append zero bytes; past the end of the file every
    // byte reads as zero; a code or a fast-path coefficient longer than the bits in the buffer is an error, magnitude bits
    // that are missing read as the value 0.
    void fill() {
        do {
            const int b = no_more ? 0 : byte();
            if (b == 0xff) {
                int c = byte();
                while (c == 0xff) c = byte();   // past the end bytes read as 0
                if (c != 0) {
                    pending_marker = c;
                    no_more = true;
                    return;
                }
            }
            bit_buffer |= (uint32_t)b << (24 - bit_count);
            bit_count += 8;
        } while (bit_count <= 24);
    }
    void drop(int n) {
        bit_buffer <<= n;
        bit_count -= n;
    }
    int bits(int n) {   // n <= 16 unsigned bits
        if (bit_count < n) fill();
        if (bit_count < n || n == 0) return 0;
        const uint32_t v = bit_buffer >> (32 - n);
        drop(n);
        return (int)v;
    }
    static int extended(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
    int extend(int n) {   // T.81 F.2.2.1: receive n bits and sign-extend
        if (bit_count < n) fill();
        if (bit_count < n || n == 0) return 0;
        return extended(bits(n), n);
    }
    // length of the code at the head of the buffer and its symbol; 0 if no code matches
    int peek_code(const HuffmanTable &t, int &sym) const {
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | (int)((bit_buffer >> (32 - len)) & 1);
            const int idx = code - t.first_code[len];
            if (idx >= 0 && idx < t.count[len]) {
                sym = t.symbols[t.first_index[len] + idx];
                return len;
            }
        }
        return 0;
    }
    int symbol(const HuffmanTable &t) {
        if (bit_count < 16) fill();
        int sym = 0;
        const int len = peek_code(t, sym);
        if (len == 0 || len > bit_count) return -1;
        drop(len);
        return sym;
    }
    // One AC code with its magnitude bits.  rs = the code's symbol, value = the sign-extended coefficient when the symbol
    // carries one.  Codes whose length plus magnitude bits fit 9 bits and whose value fits -128..127 are looked up as one
    // unit by the reference's decoder, so for them all the bits have to be in the buffer.
    bool ac_code(const HuffmanTable &t, int &rs, int &value) {
        if (bit_count < 16) fill();
        const int len = peek_code(t, rs);
        const int s = rs & 15;
        if (len != 0 && s != 0 && len + s <= 9) {
            const int v = extended((int)((bit_buffer << len) >> (32 - s)), s);
            if (v >= -128 && v <= 127) {
                if (len + s > bit_count) return fail("bad huffman code");
                drop(len + s);
                value = v;
                return true;
            }
        }
        rs = symbol(t);   // tops the buffer up once more before it decides
        if (rs < 0) return fail("bad huffman code");
        value = (rs & 15) ? extend(rs & 15) : 0;
        return true;
    }
    void reset_entropy() {
        bit_buffer = 0;
        bit_count = 0;
        no_more = false;
        pending_marker = -1;
        eob_run = 0;
        for (int i = 0; i < 4; ++i) comp[i].dc_pred = 0;
    }

    // ---- blocks ----------------------------------------------------------------------------------------------
    static bool fits_short(int a, int b) {   // a * b representable in 16 bits
        const long long p = (long long)a * b;
        return p >= -32768 && p <= 32767;
    }

    bool dc_step(Component &c, const HuffmanTable &hd, int scale, int16_t *data) {
        const int t = symbol(hd);
        if (t < 0 || t > 15) return fail("bad huffman code");
        const long long dc = (long long)c.dc_pred + (t ? extend(t) : 0);
        if (dc < INT32_MIN || dc > INT32_MAX) return fail("bad delta");
        c.dc_pred = (int)dc;
        if (!fits_short(c.dc_pred, scale)) return fail("dc out of range");
        data[0] = (int16_t)(c.dc_pred * scale);
        return true;
    }

    bool baseline_block(Component &c, int16_t *data) {
        const HuffmanTable &ha = ac_tables[c.ha];
        const uint16_t *q = quant[c.tq];
        if (bit_count < 16) fill();
        memset(data, 0, 64 * sizeof(int16_t));
        if (!dc_step(c, dc_tables[c.hd], q[0], data)) return false;
        int k = 1;
        do {
            int rs, v;
            if (!ac_code(ha, rs, v)) return false;
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                const int zig = ZIGZAG[k++];
                data[zig] = (int16_t)(v * q[zig]);
            }
        } while (k < 64);
        return true;
    }

    bool progressive_dc(Component &c, int16_t *data) {
        if (spec_end != 0) return fail("can't merge dc and ac");
        if (bit_count < 16) fill();
        if (succ_high == 0) {
            memset(data, 0, 64 * sizeof(int16_t));
            return dc_step(c, dc_tables[c.hd], 1 << succ_low, data);
        }
        if (bits(1)) data[0] = (int16_t)(data[0] + (1 << succ_low));
        return true;
    }

    bool progressive_ac(Component &c, int16_t *data) {
        if (spec_start == 0) return fail("can't merge dc and ac");
        const HuffmanTable &ha = ac_tables[c.ha];
        if (succ_high == 0) {
            if (eob_run) {
                --eob_run;
                return true;
            }
            int k = spec_start;
            do {
                int rs, v;
                if (!ac_code(ha, rs, v)) return false;
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r) + (r ? bits(r) : 0) - 1;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const int zig = ZIGZAG[k++];
                    data[zig] = (int16_t)(v * (1 << succ_low));
                }
            } while (k <= spec_end);
            return true;
        }
        // refinement of coefficients that are already non-zero, interleaved with newly non-zero ones (T.81 G.1.2.3)
        const int16_t bit = (int16_t)(1 << succ_low);
        auto refine = [&](int16_t &p) {
            if (bits(1) && (p & bit) == 0) p = (int16_t)(p > 0 ? p + bit : p - bit);
        };
        if (eob_run) {
            --eob_run;
            for (int k = spec_start; k <= spec_end; ++k) {
                int16_t &p = data[ZIGZAG[k]];
                if (p != 0) refine(p);
            }
            return true;
        }
        int k = spec_start;
        do {
            const int rs = symbol(ha);
            if (rs < 0) return fail("bad huffman code");
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1 + (r ? bits(r) : 0);
                    r = 64;   // run to the end of the band
                }
            } else {
                if (s != 1) return fail("bad huffman code");
                s = bits(1) ? bit : -bit;
            }
            while (k <= spec_end) {
                int16_t &p = data[ZIGZAG[k++]];
                if (p != 0) {
                    refine(p);
                } else {
                    if (r == 0) {
                        p = (int16_t)s;
                        break;
                    }
                    --r;
                }
            }
        } while (k <= spec_end);
        return true;
    }

    static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

    // One dimension of the inverse DCT on 16-bit inputs; outputs scaled by 4096: x[0..3] the even part, t[0..3] the odd
    // part.  The reference runs its decoder's 8-lane variant of this transform (third_party/stb/stb_image.h:2530-2703),
    // whose regrouping is followed here because it decides the result for out-of-range coefficients (truncated or damaged
    // files): s0 +- s4, s1 + s7 and s3 + s5 wrap to 16 bits, products accumulate modulo 2^32.  For coefficients a real
    // encoder produces nothing wraps and this is the plain 12-bit "islow" transform.
    static void idct_1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int *x, int *t) {
        auto fx = [](double v) { return (int)(v * 4096 + 0.5); };
        static const int c0541 = fx(0.5411961f), cm1847 = fx(-1.847759065f), c0765 = fx(0.765366865f), c1175 = fx(1.175875602f),
                         c0298 = fx(0.298631336f), c2053 = fx(2.053119869f), c3072 = fx(3.072711026f), c1501 = fx(1.501321110f),
                         cm0899 = fx(-0.899976223f), cm2562 = fx(-2.562915447f), cm1961 = fx(-1.961570560f), cm0390 = fx(-0.390180644f);
        auto dot = [](int a, int ca, int b, int cb) { return (uint32_t)(a * ca) + (uint32_t)(b * cb); };
        const uint32_t e2 = dot(s2, c0541, s6, c0541 + cm1847), e3 = dot(s2, c0541 + c0765, s6, c0541);
        const uint32_t e0 = (uint32_t)((int16_t)(s0 + s4) * 4096), e1 = (uint32_t)((int16_t)(s0 - s4) * 4096);
        x[0] = (int)(e0 + e3);
        x[3] = (int)(e0 - e3);
        x[1] = (int)(e1 + e2);
        x[2] = (int)(e1 - e2);
        const int s17 = (int16_t)(s1 + s7), s35 = (int16_t)(s3 + s5);
        const uint32_t y4 = dot(s17, c1175 + cm0899, s35, c1175), y5 = dot(s17, c1175, s35, c1175 + cm2562);
        t[0] = (int)(dot(s7, cm1961 + c0298, s3, cm1961) + y4);
        t[1] = (int)(dot(s5, cm0390 + c2053, s1, cm0390) + y5);
        t[2] = (int)(dot(s7, cm1961, s3, cm1961 + c3072) + y5);
        t[3] = (int)(dot(s5, cm0390, s1, cm0390 + c1501) + y4);
    }

    static int wrap_add(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
    static int wrap_sub(int a, int b) { return (int)((uint32_t)a - (uint32_t)b); }
    static int sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

    static void idct_block(uint8_t *out, int stride, const int16_t *d) {
        int val[64];
        for (int i = 0; i < 8; ++i) {
            const int16_t *c = d + i;
            int *v = val + i;
            int x[4], t[4];
            idct_1d(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], x, t);
            for (int k = 0; k < 4; ++k) {
                const int e = wrap_add(x[k], 512);
                v[8 * k] = sat16(wrap_add(e, t[3 - k]) >> 10);
                v[8 * (7 - k)] = sat16(wrap_sub(e, t[3 - k]) >> 10);
            }
        }
        for (int i = 0; i < 8; ++i) {
            const int *v = val + 8 * i;
            uint8_t *o = out + (size_t)stride * i;
            int x[4], t[4];
            idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x, t);
            for (int k = 0; k < 4; ++k) {
                const int e = wrap_add(x[k], 65536 + (128 << 17));
                o[k] = clamp8(wrap_add(e, t[3 - k]) >> 17);
                o[7 - k] = clamp8(wrap_sub(e, t[3 - k]) >> 17);
            }
        }
    }

    // ---- scans -------------------------------------------------------------------------------------------------
    bool decode_block_at(Component &c, int bx, int by) {
        if (bx >= c.blocks_w || by >= c.blocks_h) return fail("block outside the image");
        if (progressive) {
            int16_t *data = &c.coeff[64 * ((size_t)by * c.blocks_w + bx)];
            return spec_start == 0 || scan_n > 1 ? progressive_dc(c, data) : progressive_ac(c, data);
        }
        int16_t data[64];
        if (!baseline_block(c, data)) return false;
        idct_block(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, data);
        return true;
    }

    // after `restart_interval` MCUs: expect RSTn; false = the scan ends here
    bool restart_point(int &todo) {
        if (--todo > 0) return true;
        if (bit_count < 24) fill();
        if (!(pending_marker >= 0xd0 && pending_marker <= 0xd7)) return false;
        reset_entropy();
        todo = restart_interval ? restart_interval : 0x7fffffff;
        return true;
    }

    bool decode_scan() {
        reset_entropy();
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        if (scan_n == 1) {   // non-interleaved: the component's own blocks, row by row
            Component &c = comp[scan_order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    if (!decode_block_at(c, i, j)) return false;
                    if (!restart_point(todo)) return true;
                }
            return true;
        }
        for (int j = 0; j < mcus_y; ++j)
            for (int i = 0; i < mcus_x; ++i) {
                for (int k = 0; k < scan_n; ++k) {
                    Component &c = comp[scan_order[k]];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x)
                            if (!decode_block_at(c, i * c.h + x, j * c.v + y)) return false;
                }
                if (!restart_point(todo)) return true;
            }
        return true;
    }

    void finish_progressive() {
        for (int n = 0; n < n_comp; ++n) {
            Component &c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    int16_t *data = &c.coeff[64 * ((size_t)j * c.blocks_w + i)];
                    for (int k = 0; k < 64; ++k) data[k] = (int16_t)(data[k] * quant[c.tq][k]);
                    idct_block(&c.plane[(size_t)j * 8 * c.w2 + (size_t)i * 8], c.w2, data);
                }
        }
    }

    // ---- markers ---------------------------------------------------------------------------------------------
    int next_marker() {
        if (pending_marker >= 0) {
            const int m = pending_marker;
            pending_marker = -1;
            return m;
        }
        if (pos >= file.size()) return -1;
        int x = byte();
        if (x != 0xff) return -1;
        while (x == 0xff && pos < file.size()) x = byte();
        return x;
    }

    bool segment(int m) {
        if (m < 0) return fail("expected marker");
        if (m == 0xdd) {   // DRI
            if (word() != 4) return fail("bad DRI len");
            restart_interval = word();
            return true;
        }
        if (m == 0xdb) {   // DQT
            int len = word() - 2;
            while (len > 0) {
                const int q = byte(), wide = q >> 4, t = q & 15;
                if ((wide != 0 && wide != 1) || t > 3) return fail("bad DQT");
                for (int i = 0; i < 64; ++i) quant[t][ZIGZAG[i]] = (uint16_t)(wide ? word() : byte());
                len -= wide ? 129 : 65;
            }
            return len == 0 ? true : fail("bad DQT len");
        }
        if (m == 0xc4) {   // DHT
            int len = word() - 2;
            while (len > 0) {
                const int q = byte(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) return fail("bad DHT header");
                int counts[16], total = 0;
                for (int i = 0; i < 16; ++i) {
                    counts[i] = byte();
                    total += counts[i];
                }
                if (total > 256) return fail("bad DHT header");
                uint8_t values[256];
                for (int i = 0; i < total; ++i) values[i] = (uint8_t)byte();
                HuffmanTable &t = tc == 0 ? dc_tables[th] : ac_tables[th];
                if (!t.build(counts, values, total)) return fail("bad code lengths");
                len -= 17 + total;
            }
            return len == 0 ? true : fail("bad DHT len");
        }
        if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {   // APPn, COM
            int len = word();
            if (len < 2) return fail("bad segment len");
            len -= 2;
            if (m == 0xe0 && len >= 5) {
                static const char tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; ++i) ok &= byte() == (uint8_t)tag[i];
                len -= 5;
                if (ok) jfif = true;
            } else if (m == 0xee && len >= 12) {
                static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; ++i) ok &= byte() == (uint8_t)tag[i];
                len -= 6;
                if (ok) {
                    byte();    // version
                    word();    // flags0
                    word();    // flags1
                    adobe_transform = byte();
                    len -= 6;
                }
            }
            pos = std::min(file.size(), pos + (size_t)len);
            return true;
        }
        return fail("unknown marker");
    }

    bool frame_header() {
        const int len = word();
        if (len < 11) return fail("bad SOF len");
        if (byte() != 8) return fail("only 8-bit JPEG is supported");
        height = word();
        width = word();
        if (height == 0 || width == 0) return fail("bad image size");
        n_comp = byte();
        if (n_comp != 1 && n_comp != 3 && n_comp != 4) return fail("bad component count");
        if (len != 8 + 3 * n_comp) return fail("bad SOF len");
        static const char rgb_tag[3] = {'R', 'G', 'B'};
        int rgb_match = 0;
        for (int i = 0; i < n_comp; ++i) {
            Component &c = comp[i];
            c.id = byte();
            if (n_comp == 3 && c.id == rgb_tag[i]) ++rgb_match;
            const int q = byte();
            c.h = q >> 4;
            c.v = q & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return fail("bad sampling factors");
            c.tq = byte();
            if (c.tq > 3) return fail("bad quantisation table number");
            h_max = std::max(h_max, c.h);
            v_max = std::max(v_max, c.v);
        }
        rgb_ids = rgb_match == 3;
        for (int i = 0; i < n_comp; ++i)
            if (h_max % comp[i].h != 0 || v_max % comp[i].v != 0) return fail("bad sampling factors");
        if ((uint64_t)width * height * n_comp > 0x7fffffffull) return fail("image too large");
        mcu_w = h_max * 8;
        mcu_h = v_max * 8;
        mcus_x = (width + mcu_w - 1) / mcu_w;
        mcus_y = (height + mcu_h - 1) / mcu_h;
        for (int i = 0; i < n_comp; ++i) {
            Component &c = comp[i];
            c.x = (width * c.h + h_max - 1) / h_max;
            c.y = (height * c.v + v_max - 1) / v_max;
            c.w2 = mcus_x * c.h * 8;
            c.h2 = mcus_y * c.v * 8;
            c.blocks_w = c.w2 / 8;
            c.blocks_h = c.h2 / 8;
            c.plane.assign((size_t)c.w2 * c.h2 + 15, 0);
            if (progressive) c.coeff.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        return true;
    }

    bool scan_header() {
        const int len = word();
        scan_n = byte();
        if (scan_n < 1 || scan_n > 4 || scan_n > n_comp) return fail("bad SOS component count");
        if (len != 6 + 2 * scan_n) return fail("bad SOS len");
        for (int i = 0; i < scan_n; ++i) {
            const int id = byte(), q = byte();
            int which = -1;
            for (int k = 0; k < n_comp; ++k)
                if (comp[k].id == id) {
                    which = k;
                    break;
                }
            if (which < 0) return fail("bad SOS component");
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) return fail("bad huffman table number");
            scan_order[i] = which;
        }
        spec_start = byte();
        spec_end = byte();
        const int a = byte();
        succ_high = a >> 4;
        succ_low = a & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return fail("bad SOS");
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) return fail("bad SOS");
            spec_end = 63;
        }
        for (int i = 0; i < scan_n; ++i) {
            const Component &c = comp[scan_order[i]];
            const bool need_dc = !progressive || spec_start == 0, need_ac = !progressive || spec_start != 0;
            if (need_dc && succ_high == 0 && !dc_tables[c.hd].present) return fail("missing huffman table");
            if (need_ac && !ac_tables[c.ha].present) return fail("missing huffman table");
        }
        return true;
    }

    // Marker walk of the reference's decoder (third_party/stb/stb_image.h:3365-3447): everything up to the frame header has
    // to parse; after it, a segment that does not parse ends decoding and keeps the planes as they are -- for a
    // progressive file that means before the coefficients were transformed, i.e. all zero.
    bool run() {
        memset(quant, 0, sizeof quant);
        pos = 0;
        if (next_marker() != 0xd8) return fail("not a JPEG");
        int m = next_marker();
        while (m != 0xc0 && m != 0xc1 && m != 0xc2) {
            if (!segment(m)) return false;
            m = next_marker();
            while (m < 0) {   // padding between segments
                if (pos >= file.size()) return fail("no frame");
                m = next_marker();
            }
        }
        progressive = m == 0xc2;
        if (!frame_header()) return false;
        m = next_marker();
        while (m != 0xd9) {
            if (m == 0xda) {
                if (!scan_header()) return false;
                if (!decode_scan()) return false;
                if (pending_marker < 0) pending_marker = skip_to_marker();
                m = next_marker();
                if (m >= 0xd0 && m <= 0xd7) m = next_marker();
            } else if (m == 0xdc) {   // DNL
                const int len = word(), lines = word();
                if (len != 4) return fail("bad DNL len");
                if (lines != height) return fail("bad DNL height");
                m = next_marker();
            } else {
                if (!segment(m)) {
                    error.clear();
                    return true;
                }
                m = next_marker();
            }
        }
        if (progressive) finish_progressive();
        return true;
    }

    // the bytes between the end of a scan's data and the next marker
    int skip_to_marker() {
        while (pos < file.size()) {
            int x = byte();
            while (x == 0xff) {
                if (pos >= file.size()) return -1;
                x = byte();
                if (x != 0 && x != 0xff) return x;
            }
        }
        return -1;
    }

    // ---- upsampling and colour ---------------------------------------------------------------------------------
    static const uint8_t *upsample_row(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int hs, int vs) {
        if (hs == 1 && vs == 1) return near;
        if (hs == 1 && vs == 2) {
            for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
            return out;
        }
        if (hs == 2 && vs == 1) {
            if (w == 1) {
                out[0] = out[1] = near[0];
                return out;
            }
            out[0] = near[0];
            out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
            int i;
            for (i = 1; i < w - 1; ++i) {
                const int n = 3 * near[i] + 2;
                out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
                out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
            }
            out[2 * i] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
            out[2 * i + 1] = near[w - 1];
            return out;
        }
        if (hs == 2 && vs == 2) {
            if (w == 1) {
                out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
                return out;
            }
            int t1 = 3 * near[0] + far[0];
            out[0] = (uint8_t)((t1 + 2) >> 2);
            for (int i = 1; i < w; ++i) {
                const int t0 = t1;
                t1 = 3 * near[i] + far[i];
                out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
            }
            out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
            return out;
        }
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];
        return out;
    }

    static uint8_t blinn(uint8_t x, uint8_t y) {
        const unsigned t = x * y + 128;
        return (uint8_t)((t + (t >> 8)) >> 8);
    }

    static void ycc_to_rgb(uint8_t *out, int y, int cb, int cr) {
        auto fixed = [](float v) { return ((int)(v * 4096.0f + 0.5f)) << 8; };
        static const int k_r = fixed(1.40200f), k_g_cr = fixed(0.71414f), k_g_cb = fixed(0.34414f), k_b = fixed(1.77200f);
        const int y_fixed = (y << 20) + (1 << 19);
        cr -= 128;
        cb -= 128;
        int r = y_fixed + cr * k_r;
        int g = y_fixed + (cr * -k_g_cr) + (int)((unsigned)(cb * -k_g_cb) & 0xffff0000u);
        int b = y_fixed + cb * k_b;
        r >>= 20;
        g >>= 20;
        b >>= 20;
        out[0] = clamp8(r);
        out[1] = clamp8(g);
        out[2] = clamp8(b);
    }

    void convert(std::vector<uint8_t> &rgb) {
        rgb.assign((size_t)width * height * 3, 0);
        struct Row {
            int hs, vs, ystep, w_lores, ypos;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buf;
        } rows[4];
        for (int k = 0; k < n_comp; ++k) {
            Row &r = rows[k];
            r.hs = h_max / comp[k].h;
            r.vs = v_max / comp[k].v;
            r.ystep = r.vs >> 1;
            r.w_lores = (width + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = comp[k].plane.data();
            r.buf.assign((size_t)width + 8 + (size_t)r.hs * 2, 0);
        }
        const bool is_rgb = n_comp == 3 && (rgb_ids || (adobe_transform == 0 && !jfif));
        const uint8_t *line[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < height; ++j) {
            uint8_t *out = &rgb[(size_t)j * width * 3];
            for (int k = 0; k < n_comp; ++k) {
                Row &r = rows[k];
                const bool bottom = r.ystep >= (r.vs >> 1);
                line[k] = upsample_row(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
                }
            }
            for (int i = 0; i < width; ++i, out += 3) {
                if (n_comp == 1) {
                    out[0] = out[1] = out[2] = line[0][i];
                } else if (n_comp == 3) {
                    if (is_rgb) {
                        out[0] = line[0][i];
                        out[1] = line[1][i];
                        out[2] = line[2][i];
                    } else {
                        ycc_to_rgb(out, line[0][i], line[1][i], line[2][i]);
                    }
                } else if (adobe_transform == 0) {   // CMYK
                    const uint8_t m = line[3][i];
                    out[0] = blinn(line[0][i], m);
                    out[1] = blinn(line[1][i], m);
                    out[2] = blinn(line[2][i], m);
                } else {
                    ycc_to_rgb(out, line[0][i], line[1][i], line[2][i]);
                    if (adobe_transform == 2) {   // YCCK
                        const uint8_t m = line[3][i];
                        out[0] = blinn((uint8_t)(255 - out[0]), m);
                        out[1] = blinn((uint8_t)(255 - out[1]), m);
                        out[2] = blinn((uint8_t)(255 - out[2]), m);
                    }
                }
            }
        }
    }
};

}  // namespace

bool decode_jpeg(const std::vector<uint8_t> &file, int &w, int &h, std::vector<uint8_t> &rgb, std::string &err) {
    Decoder d(file);
    return d.decode(w, h, rgb, err);
}

}  // namespace rth
