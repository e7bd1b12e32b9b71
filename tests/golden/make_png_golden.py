"""Generates tests/golden/png.npz: small PNG files and the texels the reference's own decoder produces for them.

Expected bytes come from oracle/_ref/stb_dump (the reference's vendored decoder, called as in reference
src/texture.cpp:46).  The files are synthetic, written by the encoder below so that every colour type, bit depth, row
filter and both interlace methods occur, plus palette / tRNS cases.

Run from the repo root, in the container that has /root/reference:   python tests/golden/make_png_golden.py
"""
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
STB = os.path.join(ROOT, "oracle", "_ref", "stb_dump")
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}
ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def chunk(kind, data):
    body = kind + bytes(data)
    return len(data).to_bytes(4, "big") + body + zlib.crc32(body).to_bytes(4, "big")


def pack_rows(samples, depth):
    """samples: (rows, n) integer array -> list of packed byte rows."""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(r.astype(np.uint8)))
        elif depth == 16:
            rows.append(r.astype(">u2").tobytes())
        else:
            bits = np.zeros(-(-len(r) * depth // 8) * 8, dtype=np.uint8)
            for k in range(depth):
                bits[k:len(r) * depth:depth] = (r >> (depth - 1 - k)) & 1
            rows.append(bytes(np.packbits(bits)))
    return rows


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp, filters):
    out = bytearray()
    prev = None
    for y, row in enumerate(rows):
        f = filters[y % len(filters)]
        out.append(f)
        for x, v in enumerate(row):
            a = row[x - bpp] if x >= bpp else 0
            b = prev[x] if prev is not None else 0
            c = prev[x - bpp] if prev is not None and x >= bpp else 0
            pred = [0, a, b, (a + b) // 2, paeth(a, b, c)][f]
            out.append((v - pred) & 0xff)
        prev = row
    return out


def encode_png(samples, ctype, depth, interlace=False, filters=(0, 1, 2, 3, 4), palette=None, trns=None, idat_split=0):
    """samples: (H, W, channels) integers in 0 .. 2^depth - 1."""
    h, w, ch = samples.shape
    assert ch == CHANNELS[ctype]
    bpp = max(1, ch * depth // 8)
    raw = bytearray()
    for x0, y0, dx, dy in (ADAM7 if interlace else [(0, 0, 1, 1)]):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        raw += filter_rows(pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, filters)
    out = bytearray(b"\x89PNG\r\n\x1a\n")
    out += chunk(b"IHDR", w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([depth, ctype, 0, 0, 1 if interlace else 0]))
    out += chunk(b"gAMA", (45455).to_bytes(4, "big"))
    if palette is not None:
        out += chunk(b"PLTE", bytes(np.asarray(palette, dtype=np.uint8).reshape(-1)))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    z = zlib.compress(bytes(raw), 6)
    if idat_split:
        for i in range(0, len(z), idat_split):
            out += chunk(b"IDAT", z[i:i + idat_split])
    else:
        out += chunk(b"IDAT", z)
    out += chunk(b"tEXt", b"Comment\x00synthetic")
    out += chunk(b"IEND", b"")
    return bytes(out)


def make_cases():
    rng = np.random.default_rng(77)
    cases = {}

    def img(h, w, ch, depth):
        y, x = np.mgrid[0:h, 0:w]
        top = (1 << depth) - 1
        planes = [((np.sin(x * (0.2 + 0.1 * c)) * np.cos(y * 0.17 + c) * 0.5 + 0.5) * top + rng.integers(0, max(2, top // 6), (h, w)))
                  for c in range(ch)]
        return np.clip(np.stack(planes, axis=-1), 0, top).astype(np.int64)

    for interlace in (False, True):
        tag = "_adam7" if interlace else ""
        for depth in (1, 2, 4, 8, 16):
            cases[f"grey{depth}{tag}"] = encode_png(img(21, 35, 1, depth), 0, depth, interlace)
        for depth in (8, 16):
            cases[f"rgb{depth}{tag}"] = encode_png(img(19, 30, 3, depth), 2, depth, interlace)
            cases[f"grey_alpha{depth}{tag}"] = encode_png(img(17, 26, 2, depth), 4, depth, interlace)
            cases[f"rgba{depth}{tag}"] = encode_png(img(23, 29, 4, depth), 6, depth, interlace)
        for depth in (1, 2, 4, 8):
            n = 1 << depth
            pal = rng.integers(0, 256, (n, 3))
            cases[f"palette{depth}{tag}"] = encode_png(rng.integers(0, n, (18, 27, 1)), 3, depth, interlace, palette=pal)
    pal = rng.integers(0, 256, (40, 3))
    cases["palette8_trns"] = encode_png(rng.integers(0, 40, (15, 22, 1)), 3, 8, palette=pal, trns=[int(v) for v in rng.integers(0, 256, 17)])
    cases["grey8_trns"] = encode_png(img(12, 20, 1, 8), 0, 8, trns=(0).to_bytes(1, "big") + bytes([90]))
    cases["grey2_trns"] = encode_png(img(12, 20, 1, 2), 0, 2, trns=bytes([0, 1]))
    cases["rgb8_trns"] = encode_png(img(12, 20, 3, 8), 2, 8, trns=bytes([0, 10, 0, 20, 0, 30]))
    cases["rgb16_trns"] = encode_png(img(12, 20, 3, 16), 2, 16, True, trns=bytes([1, 10, 2, 20, 3, 30]))
    for f in range(5):
        cases[f"rgb8_filter{f}"] = encode_png(img(14, 18, 3, 8), 2, 8, filters=(f,))
        cases[f"grey4_filter{f}_adam7"] = encode_png(img(14, 19, 1, 4), 0, 4, True, filters=(f,))
    cases["rgb8_split_idat"] = encode_png(img(40, 40, 3, 8), 2, 8, idat_split=97)
    for w, h in ((1, 1), (1, 9), (9, 1), (2, 2), (3, 5), (8, 8), (5, 3)):
        cases[f"rgb8_{w}x{h}_adam7"] = encode_png(img(h, w, 3, 8), 2, 8, True)
        cases[f"grey1_{w}x{h}_adam7"] = encode_png(img(h, w, 1, 1), 0, 1, True)
    return cases


def stb_decode(data, tmp):
    src, dst = os.path.join(tmp, "in.png"), os.path.join(tmp, "out.bin")
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
    out, names = {}, []
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in make_cases().items():
            pixels, why = stb_decode(data, tmp)
            if pixels is None:
                print(f"{name}: reference decoder rejects it ({why}); left out")
                continue
            names.append(name)
            out["file_" + name] = np.frombuffer(data, dtype=np.uint8)
            out["rgb_" + name] = pixels
    print(f"{len(names)} files")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "png.npz"), **out)


if __name__ == "__main__":
    main()
