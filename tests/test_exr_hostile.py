"""Truncated and hostile EXR files (textures and environment maps come from scene authors): the reader of the host
library must reject or decode them without reading or writing outside its buffers.  The files are mutations of three
valid ones (uncompressed, ZIPS, ZIP blocks of 16 lines); they go through the shared library, and through a build of the
same sources with AddressSanitizer and UBSan, which is what turns a silent overrun into a failure."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import raytracer_amd as R
from golden import scenes as GS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "raytracer_amd", "csrc", "host")


def exr_bytes(rgb, compression):
    """A scan-line EXR (32-bit float B, G, R) with compression 0 (none), 2 (ZIPS) or 3 (ZIP, 16 lines per block)."""
    rgb = np.ascontiguousarray(rgb, dtype="<f4")
    h, w, _ = rgb.shape
    out = bytearray(b"\x76\x2f\x31\x01\x02\x00\x00\x00")

    def attr(name, typ, data):
        out.extend(name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data)

    ch = b"".join(c.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for c in "BGR")
    attr("channels", "chlist", ch + b"\0")
    attr("compression", "compression", bytes([compression]))
    attr("dataWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    attr("lineOrder", "lineOrder", b"\0")
    attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
    attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    out.append(0)
    per_block = 16 if compression == 3 else 1
    blocks = []
    for y0 in range(0, h, per_block):
        raw = b"".join(rgb[y, :, c].tobytes() for y in range(y0, min(h, y0 + per_block)) for c in (2, 1, 0))
        data = raw
        if compression:
            a = np.frombuffer(raw, dtype=np.uint8)
            inter = np.concatenate([a[0::2], a[1::2]]).astype(np.int16)
            inter[1:] = (inter[1:] - inter[:-1] + 128 + 256) % 256
            packed = zlib.compress(inter.astype(np.uint8).tobytes(), 6)
            data = packed if len(packed) < len(raw) else raw
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    table = len(out)
    at = table + 8 * len(blocks)
    for b in blocks:
        out.extend(struct.pack("<Q", at))
        at += len(b)
    for b in blocks:
        out.extend(b)
    return bytes(out)


def hostile_files(tmp):
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(7)
    img = GS.sky_image(24, 20)
    paths = []

    def put(name, data):
        path = os.path.join(tmp, name)
        with open(path, "wb") as f:
            f.write(data)
        paths.append(path)

    for comp in (0, 2, 3):
        good = exr_bytes(img, comp)
        put(f"good_{comp}.exr", good)
        for cut in list(range(0, 400, 7)) + [len(good) - 1, len(good) - 13, len(good) // 2]:
            put(f"cut_{comp}_{cut}.exr", good[:max(0, cut)])
        header_end = good.index(b"screenWindowWidth") + 30
        for k in range(120):                                     # random bytes in the header and the offset table
            b = bytearray(good)
            for _ in range(1 + k % 3):
                b[rng.integers(8, header_end + 8 * 20)] = rng.integers(0, 256)
            put(f"header_{comp}_{k}.exr", bytes(b))
        dw = good.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
        for k, window in enumerate([(0, 0, 2**31 - 1, 2**31 - 1), (-2**31, -2**31, 2**31 - 1, 5), (0, 5, 23, 2), (0, 0, 65535, 65535),
                                    (0, -4, 23, 19), (0, 0, 23, 40), (5, 0, 2, 19)]):
            b = bytearray(good)
            b[dw:dw + 16] = struct.pack("<iiii", *window)
            put(f"window_{comp}_{k}.exr", bytes(b))
        table = good.index(b"screenWindowWidth\0float\0") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1
        first_block = struct.unpack_from("<Q", good, table)[0]
        for k, (y0, dsize) in enumerate([(-5, None), (19, None), (400, None), (2**31 - 1, None), (None, -1), (None, -2**31), (None, 2**31 - 1),
                                         (None, 3), (None, 0)]):
            b = bytearray(good)
            if y0 is not None:
                struct.pack_into("<i", b, first_block, y0)
            if dsize is not None:
                struct.pack_into("<i", b, first_block + 4, dsize)
            put(f"block_{comp}_{k}.exr", bytes(b))
        for k, off in enumerate([0, 7, len(good) - 4, len(good), 2**40, 2**64 - 1, 2**64 - 8]):
            b = bytearray(good)
            struct.pack_into("<Q", b, table, off)
            put(f"offset_{comp}_{k}.exr", bytes(b))
        ct = good.index(b"B\0") + 2
        for k, t in enumerate([3, -1, 255, 1, 0]):                # channel types: unknown ones, then HALF / UINT with FLOAT-sized data
            b = bytearray(good)
            struct.pack_into("<i", b, ct, t)
            put(f"chantype_{comp}_{k}.exr", bytes(b))
    return paths


def test_valid_files_decode(tmp_path):
    h = R.host_lib()
    img = GS.sky_image(24, 20)
    for comp in (0, 2, 3):
        path = str(tmp_path / f"g{comp}.exr")
        open(path, "wb").write(exr_bytes(img, comp))
        w_, h_ = C.c_int32(), C.c_int32()
        rgba = np.empty((20, 24, 4), dtype=np.float32)
        assert h.rth_read_exr(os.fsencode(path), C.byref(w_), C.byref(h_), rgba.ctypes.data) == 0, h.rth_last_error()
        assert (w_.value, h_.value) == (24, 20) and np.array_equal(rgba[:, :, :3], img) and (rgba[:, :, 3] == 1).all()


def test_hostile_files_through_the_library(tmp_path):
    h = R.host_lib()
    ok = 0
    for path in hostile_files(str(tmp_path / "files")):
        w_, h_ = C.c_int32(), C.c_int32()
        rc = h.rth_read_exr(os.fsencode(path), C.byref(w_), C.byref(h_), None)
        assert rc in (0, 1)
        if rc == 0:
            ok += 1
            assert 0 < w_.value <= 65536 and 0 < h_.value <= 65536
    assert ok >= 3                                                # the three valid files at least


def test_hostile_files_under_address_sanitizer(tmp_path):
    exe = str(tmp_path / "exr_probe")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", f"-I{HOST}", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "tools", "exr_probe.cpp"), os.path.join(HOST, "image_io.cpp"), os.path.join(HOST, "jpeg.cpp"),
                            "-lz", "-o", exe], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    files = hostile_files(str(tmp_path / "files"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048")
    env.pop("LD_PRELOAD", None)                                   # the sanitizer runtime has to come first in the probe
    for k in range(0, len(files), 100):
        run = subprocess.run([exe] + files[k:k + 100], capture_output=True, text=True, timeout=300, env=env)
        assert run.returncode == 0, run.stderr[-3000:]
        assert run.stdout.count("\n") == len(files[k:k + 100])
