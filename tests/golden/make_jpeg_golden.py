"""Generates tests/golden/jpeg.npz: small JPEG files and the texels the reference's own decoder produces for them.

The expected bytes come from oracle/_ref/stb_dump, which is the reference's vendored 8-bit image decoder compiled from
where it lies and called as the reference calls it (reference src/texture.cpp:46).  The JPEG files themselves are
synthetic: Pillow writes the common kinds (baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, CMYK, restart markers,
optimised tables) and the small baseline encoder below writes the kinds Pillow cannot (1x2, 4x1, 2x4 ... sampling, RGB
component ids, Adobe transform 0, restart intervals inside non-interleaved scans, 16-bit quantisation tables).

Run from the repo root, in the container that has /root/reference:   python tests/golden/make_jpeg_golden.py
"""
import io
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
STB = os.path.join(ROOT, "oracle", "_ref", "stb_dump")

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63])

# Huffman tables of T.81 annex K.3 (luminance DC / AC); used for every component here
DC_COUNTS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
DC_VALUES = list(range(12))
AC_COUNTS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d]
AC_VALUES = [
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa]


def huffman_codes(counts, values):
    codes, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            codes[values[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return codes


DC_CODES = huffman_codes(DC_COUNTS, DC_VALUES)
AC_CODES = huffman_codes(AC_COUNTS, AC_VALUES)


class BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xff
            self.out.append(b)
            if b == 0xff:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def dct_matrix():
    k = np.arange(8)
    m = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * 0.5
    m[0] *= 1 / np.sqrt(2)
    return m


DCT = dct_matrix()


def magnitude(v):
    a = abs(int(v))
    size = a.bit_length()
    return size, (v if v >= 0 else v + (1 << size) - 1)


def encode_block(bw, coeffs, pred):
    """coeffs: 64 quantised values in zigzag order."""
    diff = int(coeffs[0]) - pred
    size, bits = magnitude(diff)
    bw.put(*DC_CODES[size])
    if size:
        bw.put(bits, size)
    run = 0
    last = max([i for i in range(1, 64) if coeffs[i] != 0], default=0)
    for i in range(1, last + 1):
        v = int(coeffs[i])
        if v == 0:
            run += 1
            continue
        while run > 15:
            bw.put(*AC_CODES[0xf0])
            run -= 16
        size, bits = magnitude(v)
        bw.put(*AC_CODES[(run << 4) | size])
        bw.put(bits, size)
        run = 0
    if last < 63:
        bw.put(*AC_CODES[0x00])
    return int(coeffs[0])


def segment(marker, payload):
    return bytes([0xff, marker]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)


def encode_jpeg(planes, sampling, ids=None, quant_scale=1, interleaved=True, restart=0, app=b"", wide_quant=False):
    """Baseline JPEG of full-resolution component planes (each H x W, uint8).  sampling = [(h, v), ...]; each component
    is box-filtered down to its own resolution.  One quantisation table (flat ramp) shared by all components."""
    height, width = planes[0].shape
    n = len(planes)
    ids = ids or list(range(1, n + 1))
    h_max = max(h for h, _ in sampling)
    v_max = max(v for _, v in sampling)
    mcus_x = -(-width // (8 * h_max))
    mcus_y = -(-height // (8 * v_max))
    quant = np.clip((np.add.outer(np.arange(8), np.arange(8)) * 2 + 3) * quant_scale, 1, 65535 if wide_quant else 255).astype(np.int64)
    blocks = []
    for p, (h, v) in zip(planes, sampling):
        fx, fy = h_max // h, v_max // v
        cw, ch = -(-width * h // h_max), -(-height * v // v_max)
        full = np.pad(p.astype(np.float64), ((0, ch * fy - height), (0, cw * fx - width)), mode="edge")
        low = full.reshape(ch, fy, cw, fx).mean(axis=(1, 3))
        bw_, bh_ = mcus_x * h, mcus_y * v
        low = np.pad(low, ((0, bh_ * 8 - ch), (0, bw_ * 8 - cw)), mode="edge") - 128.0
        tiles = low.reshape(bh_, 8, bw_, 8).transpose(0, 2, 1, 3)
        freq = DCT @ tiles @ DCT.T
        q = np.rint(freq / quant).astype(np.int64).reshape(bh_, bw_, 64)[:, :, ZIGZAG]
        blocks.append((q, cw, ch))
    out = bytearray(b"\xff\xd8") + app
    qz = quant.reshape(64)[ZIGZAG]
    if wide_quant:
        out += segment(0xdb, bytes([0x10]) + b"".join(int(x).to_bytes(2, "big") for x in qz))
    else:
        out += segment(0xdb, bytes([0x00]) + bytes(int(x) for x in qz))
    sof = bytes([8]) + height.to_bytes(2, "big") + width.to_bytes(2, "big") + bytes([n])
    for cid, (h, v) in zip(ids, sampling):
        sof += bytes([cid, (h << 4) | v, 0])
    out += segment(0xc0, sof)
    out += segment(0xc4, bytes([0x00]) + bytes(DC_COUNTS) + bytes(DC_VALUES))
    out += segment(0xc4, bytes([0x10]) + bytes(AC_COUNTS) + bytes(AC_VALUES))
    if restart:
        out += segment(0xdd, restart.to_bytes(2, "big"))

    def scan(components, units):
        nonlocal out
        sos = bytes([len(components)])
        for c in components:
            sos += bytes([ids[c], 0x00])
        out += segment(0xda, sos + bytes([0, 63, 0]))
        bw = BitWriter()
        pred = [0] * n
        count, rst = 0, 0
        for unit in units:
            for c, by, bx in unit:
                pred[c] = encode_block(bw, blocks[c][0][by, bx], pred[c])
            count += 1
            if restart and count % restart == 0 and count != len(units):
                bw.flush()
                bw.out += bytes([0xff, 0xd0 + (rst & 7)])
                rst += 1
                pred = [0] * n
        bw.flush()
        out += bw.out

    if interleaved and n > 1:
        units = []
        for my in range(mcus_y):
            for mx in range(mcus_x):
                units.append([(c, my * v + y, mx * h + x) for c, (h, v) in enumerate(sampling) for y in range(v) for x in range(h)])
        scan(list(range(n)), units)
    else:
        for c in range(n):
            _, cw, ch = blocks[c]
            scan([c], [[(c, by, bx)] for by in range(-(-ch // 8)) for bx in range(-(-cw // 8))])
    out += b"\xff\xd9"
    return bytes(out)


def test_image(rng, w, h, channels):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    planes = []
    for c in range(channels):
        a, b, ph = rng.uniform(0.05, 0.6, 3)
        p = 128 + 90 * np.sin(a * x + ph * 5) * np.cos(b * y + c) + rng.normal(0, 12, (h, w))
        p[h // 3:h // 2, w // 4:w // 2] = rng.integers(0, 256)          # a hard-edged patch
        planes.append(np.clip(p, 0, 255).astype(np.uint8))
    return planes


def pil_bytes(img, **kw):
    buf = io.BytesIO()
    img.save(buf, "JPEG", **kw)
    return buf.getvalue()


def adobe(transform):
    return segment(0xee, b"Adobe\x00" + bytes([100, 0, 0, 0, 0, transform]))


JFIF = segment(0xe0, b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")


def make_cases():
    rng = np.random.default_rng(20240611)
    cases = {}
    rgb = Image.fromarray(np.stack(test_image(rng, 61, 45, 3), axis=-1), "RGB")
    big = Image.fromarray(np.stack(test_image(rng, 130, 67, 3), axis=-1), "RGB")
    cases["pil_444"] = pil_bytes(rgb, quality=90, subsampling=0)
    cases["pil_422"] = pil_bytes(rgb, quality=75, subsampling=1)
    cases["pil_420"] = pil_bytes(rgb, quality=60, subsampling=2)
    cases["pil_420_optimized"] = pil_bytes(big, quality=85, subsampling=2, optimize=True)
    cases["pil_progressive_444"] = pil_bytes(rgb, quality=90, subsampling=0, progressive=True)
    cases["pil_progressive_420"] = pil_bytes(big, quality=70, subsampling=2, progressive=True)
    cases["pil_progressive_422_q30"] = pil_bytes(big, quality=30, subsampling=1, progressive=True)
    cases["pil_grey"] = pil_bytes(rgb.convert("L"), quality=80)
    cases["pil_grey_progressive"] = pil_bytes(big.convert("L"), quality=50, progressive=True)
    cases["pil_cmyk"] = pil_bytes(rgb.convert("CMYK"), quality=85)
    cases["pil_restart_420"] = pil_bytes(big, quality=80, subsampling=2, restart_marker_blocks=3)
    cases["pil_restart_rows_progressive"] = pil_bytes(big, quality=80, subsampling=1, progressive=True, restart_marker_rows=1)
    cases["pil_1x1"] = pil_bytes(rgb.crop((5, 5, 6, 6)), quality=90, subsampling=2)
    cases["pil_2x3_422"] = pil_bytes(rgb.crop((5, 5, 7, 8)), quality=90, subsampling=1)
    cases["pil_17x1_420"] = pil_bytes(rgb.crop((0, 0, 17, 1)), quality=90, subsampling=2)
    cases["pil_q100"] = pil_bytes(rgb, quality=100, subsampling=0)
    cases["pil_q5"] = pil_bytes(rgb, quality=5, subsampling=2)
    # a file cut off in the middle of its scan keeps what was decoded
    cases["pil_truncated"] = cases["pil_420_optimized"][: len(cases["pil_420_optimized"]) * 2 // 3]

    p3 = test_image(rng, 53, 38, 3)
    p4 = test_image(rng, 37, 29, 4)
    cases["own_h1v2"] = encode_jpeg(p3, [(1, 2), (1, 1), (1, 1)], app=JFIF)
    cases["own_h4v1"] = encode_jpeg(p3, [(4, 1), (1, 1), (1, 1)], app=JFIF)
    cases["own_h2v4_mixed"] = encode_jpeg(p3, [(2, 4), (1, 2), (2, 1)], app=JFIF)
    cases["own_h4v4_h2v2"] = encode_jpeg(p3, [(4, 4), (2, 2), (1, 1)], app=JFIF)
    cases["own_chroma_finer"] = encode_jpeg(p3, [(1, 1), (2, 2), (2, 1)], app=JFIF)
    cases["own_rgb_ids"] = encode_jpeg(p3, [(1, 1)] * 3, ids=[ord("R"), ord("G"), ord("B")])
    cases["own_adobe_rgb"] = encode_jpeg(p3, [(2, 2), (1, 1), (1, 1)], app=adobe(0))
    cases["own_adobe_rgb_with_jfif"] = encode_jpeg(p3, [(1, 1)] * 3, app=JFIF + adobe(0))
    cases["own_adobe_ycc"] = encode_jpeg(p3, [(2, 1), (1, 1), (1, 1)], app=adobe(1))
    cases["own_cmyk"] = encode_jpeg(p4, [(1, 1)] * 4, app=adobe(0))
    cases["own_ycck"] = encode_jpeg(p4, [(2, 2), (1, 1), (1, 1), (2, 2)], app=adobe(2))
    cases["own_four_no_adobe"] = encode_jpeg(p4, [(1, 1)] * 4)
    cases["own_noninterleaved_420"] = encode_jpeg(p3, [(2, 2), (1, 1), (1, 1)], interleaved=False, app=JFIF)
    cases["own_noninterleaved_restart"] = encode_jpeg(p3, [(2, 1), (1, 1), (1, 1)], interleaved=False, restart=5, app=JFIF)
    cases["own_restart_1"] = encode_jpeg(p3, [(2, 2), (1, 1), (1, 1)], restart=1, app=JFIF)
    cases["own_restart_9_wraps"] = encode_jpeg(test_image(rng, 96, 80, 3), [(1, 1)] * 3, restart=9, app=JFIF)
    cases["own_wide_quant"] = encode_jpeg(p3, [(2, 2), (1, 1), (1, 1)], quant_scale=9, wide_quant=True, app=JFIF)
    cases["own_coarse_quant_clamps"] = encode_jpeg([np.where(p > 128, 255, 0).astype(np.uint8) for p in p3], [(2, 2), (1, 1), (1, 1)],
                                                   quant_scale=4, app=JFIF)
    cases["own_grey_7x9"] = encode_jpeg([p3[0][:9, :7]], [(1, 1)])
    cases["own_grey_h2v2"] = encode_jpeg([p3[0]], [(2, 2)])
    # damaged files: what is kept and what is rejected follows the reference's decoder
    def patched(data, at, value):
        out = bytearray(data)
        out[at] = value
        return bytes(out)

    base = cases["own_h2v4_mixed"]
    scan = base.index(b"\xff\xda")
    cases["damaged_marker_inside_scan"] = patched(base, scan + (len(base) - scan) * 4 // 5, 0xff)
    cases["damaged_bits_inside_scan"] = patched(base, scan + (len(base) - scan) // 2, 0x5a)
    prog = cases["pil_progressive_420"]
    cases["damaged_progressive_truncated"] = prog[: len(prog) * 3 // 4]
    sos = prog.index(b"\xff\xda")
    cases["damaged_unknown_segment_after_frame"] = prog[:sos] + b"\xff\xc8\x00\x02" + prog[sos:]
    ok = cases["pil_restart_420"]
    cases["junk_before_eoi"] = ok[:-2] + bytes([0x12, 0x00, 0xff, 0x00, 0x34, 0xff, 0xff]) + ok[-2:]
    cases["no_eoi"] = ok[:-2]
    cases["dnl_segment"] = ok[:-2] + b"\xff\xdc\x00\x04" + (67).to_bytes(2, "big") + ok[-2:]
    return cases


def stb_decode(data, tmp):
    src, dst = os.path.join(tmp, "in.jpg"), os.path.join(tmp, "out.bin")
    with open(src, "wb") as f:
        f.write(data)
    r = subprocess.run([STB, src, dst], capture_output=True, text=True)
    raw = np.fromfile(dst, dtype=np.uint8)
    w, h = (int(v) for v in raw[:8].view(np.int32))
    if r.returncode != 0:
        return None, r.stderr.strip()
    return raw[8:].reshape(h, w, 3).copy(), ""


def main():
    if not os.path.exists(STB):
        sys.exit("build oracle/_ref/stb_dump first (make -C oracle)")
    out = {}
    names = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in make_cases().items():
            pixels, why = stb_decode(data, tmp)
            names.append(name)
            out["file_" + name] = np.frombuffer(data, dtype=np.uint8)
            if pixels is None:
                print(f"{name}: reference decoder rejects it ({why})")
                out["rgb_" + name] = np.zeros((0, 0, 3), dtype=np.uint8)
            else:
                print(f"{name}: {pixels.shape[1]}x{pixels.shape[0]}, {len(data)} bytes")
                out["rgb_" + name] = pixels
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "jpeg.npz"), **out)


if __name__ == "__main__":
    main()
