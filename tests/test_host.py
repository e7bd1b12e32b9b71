"""Host-side logic (no GPU): the C ABI surface, scene ingest, tone mapping and image files.

Reference counterparts: apply_tonemap and the PNG/EXR writers (src/main.cpp:169-345, 561-607), the CLI argument
checks (src/main.cpp:642-703).  tests/golden/outputs.npz holds files the compiled reference wrote
(tests/golden/make_output_golden.py)."""
import ctypes as C
import os
import re
import subprocess
import zlib

import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "outputs.npz"))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z_0-9]+)\s*\(", text)))


def test_gpu_library_exports_every_declared_symbol():
    names = declared_functions("gpu_render.h")
    assert {"rt_scene_create", "rt_render", "rt_render_shard", "rt_scene_destroy", "rt_last_error"} <= set(names)
    lib = R.gpu_lib()
    for n in names:
        assert hasattr(lib, n), n


def test_host_library_exports_every_declared_symbol():
    names = [n for n in declared_functions("rt_host.h") if n.startswith("rth_")]
    assert len(names) >= 14
    lib = R.host_lib()
    for n in names:
        assert hasattr(lib, n), n


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof() of every ABI struct as compiled by gcc equals the ctypes mirror."""
    pairs = {"rt_material": A.Material, "rt_instance": A.Instance, "rt_triangle": A.Triangle, "rt_sphere": A.Sphere,
             "rt_plane": A.Plane, "rt_image": A.Image, "rt_texture": A.Texture, "rt_light": A.Light,
             "rt_bvh_node": A.BvhNode, "rt_bvh_prim": A.BvhPrim, "rt_scene_desc": A.SceneDesc,
             "rt_camera_desc": A.CameraDesc, "rt_stats": A.Stats}
    src = tmp_path / "sizes.c"
    body = "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in pairs)
    src.write_text('#include <stdio.h>\n#include "gpu_render.h"\nint main(void){' + body + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        name, size = line.split()
        assert C.sizeof(pairs[name]) == int(size), name


def test_missing_gpu_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(R, "_gpu", None)
    monkeypatch.setattr(R, "_HERE", str(tmp_path))
    with pytest.raises(R.RaytracerError, match="no CPU fallback"):
        R.gpu_lib()


def test_ingest_simple_scene():
    sc = R.Scene(os.path.join(ROOT, "scenes", "simple.json"))
    d = sc.desc
    assert sc.n_cameras == 1 and d.abi_version == 3
    cam = sc.camera(0)
    assert (cam.resx, cam.resy, cam.num_samples) == (64, 64, 1)
    assert d.n_bvh_prims == d.n_triangles + d.n_spheres and d.n_bvh_nodes >= 1
    bounds, meta, prims = sc.bvh_arrays()
    assert (bounds[:, :3] <= bounds[:, 3:]).all()
    leaves = meta[:, 1] > 0
    assert meta[leaves, 1].sum() == d.n_bvh_prims          # every primitive is in exactly one leaf
    assert sorted(map(tuple, prims)) == sorted(set(map(tuple, prims)))
    with pytest.raises(R.RaytracerError):
        R.Scene(os.path.join(ROOT, "scenes", "does_not_exist.json"))


@pytest.mark.parametrize("index,name", [(0, "tm_phot"), (1, "tm_filmic"), (2, "tm_aces")])
def test_tonemap_matches_reference_files(index, name):
    """Our tone mapper on the EXR the reference wrote reproduces the PNG the reference wrote.  The reference is
    built with -ffast-math (powf/logf differ in the last ulp), so a value may land on the other side of a rounding
    boundary: off-by-one on at most 0.5 % of the values, never more than one level."""
    hdr = np.ascontiguousarray(GOLD["hdr"])
    tmo, o1, o2, sat, gamma = GOLD["tm_params"][index]
    out = np.empty(hdr.shape, dtype=np.uint8)
    h, w = hdr.shape[:2]
    assert R.host_lib().rth_tonemap(int(tmo), float(o1), float(o2), float(sat), float(gamma), w, h, hdr.ctypes.data,
                                    out.ctypes.data) == 0
    diff = np.abs(out.astype(np.int32) - GOLD[name].astype(np.int32))
    assert diff.max() <= 1
    assert (diff != 0).mean() <= 0.005
    assert GOLD[name].std() > 10       # the fixture is a real image, not a constant


def test_ldr_quantisation_matches_reference_png():
    hdr = np.ascontiguousarray(GOLD["ldr_hdr"])
    out = np.empty(hdr.shape, dtype=np.uint8)
    R.host_lib().rth_quantize_ldr(hdr.size, hdr.ctypes.data, out.ctypes.data)
    assert np.array_equal(out, GOLD["ldr_png"])


def png_pixels(path):
    """Minimal PNG reader for 8-bit RGB files written without interlace."""
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(raw):
        n = int.from_bytes(raw[pos:pos + 4], "big")
        kind = raw[pos + 4:pos + 8]
        data = raw[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + data) == int.from_bytes(raw[pos + 8 + n:pos + 12 + n], "big")
        if kind == b"IHDR":
            w, h = int.from_bytes(data[:4], "big"), int.from_bytes(data[4:8], "big")
            assert data[8:10] == b"\x08\x02"
        elif kind == b"IDAT":
            idat += data
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w).astype(np.int32)
    out = np.zeros((h, 3 * w), dtype=np.int32)
    for y in range(h):
        f, line = rows[y, 0], rows[y, 1:]
        prev = out[y - 1] if y else np.zeros(3 * w, dtype=np.int32)
        if f == 0:
            out[y] = line
        elif f == 2:
            out[y] = (line + prev) & 255
        else:
            for x in range(3 * w):
                a = out[y, x - 3] if x >= 3 else 0
                c = prev[x - 3] if x >= 3 else 0
                b = prev[x]
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) // 2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                out[y, x] = (line[x] + p) & 255
    return out.reshape(h, w, 3).astype(np.uint8)


def test_png_and_exr_files_round_trip(tmp_path):
    h = R.host_lib()
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "a.png")
    assert h.rth_write_png(os.fsencode(p), 53, 37, rgb.ctypes.data, 1) == 0
    assert np.array_equal(png_pixels(p), rgb[::-1])       # flipped on write, like stbi_flip_vertically_on_write
    assert h.rth_write_png(os.fsencode(p), 53, 37, rgb.ctypes.data, 0) == 0
    assert np.array_equal(png_pixels(p), rgb)

    hdr = rng.standard_normal((19, 31, 3)).astype(np.float32) * 100
    e = str(tmp_path / "a.exr")
    assert h.rth_write_exr(os.fsencode(e), 31, 19, hdr.ctypes.data) == 0
    w, hh = C.c_int32(), C.c_int32()
    assert h.rth_read_exr(os.fsencode(e), C.byref(w), C.byref(hh), None) == 0 and (w.value, hh.value) == (31, 19)
    rgba = np.empty((19, 31, 4), dtype=np.float32)
    assert h.rth_read_exr(os.fsencode(e), C.byref(w), C.byref(hh), rgba.ctypes.data) == 0
    assert np.array_equal(rgba[:, :, :3], hdr)
    assert h.rth_read_exr(os.fsencode(str(tmp_path / "missing.exr")), C.byref(w), C.byref(hh), None) != 0


CLI = os.path.join(ROOT, "raytracer")


@pytest.mark.parametrize("args,code,text", [
    (["scene.json", "9"], 256 - 3, "Invalid threading type"),
    (["scene.json", "0", "0"], 256 - 1, "Invalid thread count"),
    ([os.path.join(ROOT, "scenes", "simple.json"), "0", "5000"], 256 - 2, ""),
    (["/nonexistent/path.json"], 256 - 1, "cannot open"),
    (["not_a_scene.txt"], 1, "neither a .json file nor a directory"),
])
def test_cli_argument_validation(args, code, text, tmp_path):
    """Same messages and exit codes as the reference's main() (src/main.cpp:642-703); none of these reach the GPU."""
    if not os.path.exists(CLI):
        pytest.skip("CLI not built")
    r = subprocess.run([CLI] + args, cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == code, (r.stdout, r.stderr)
    assert text in r.stdout + r.stderr


def _decode_image(data):
    h = R.host_lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    w, hh = C.c_int32(), C.c_int32()
    if h.rth_decode_image_rgb8(buf.ctypes.data, buf.size, C.byref(w), C.byref(hh), None) != 0:
        return None
    rgb = np.empty((hh.value, w.value, 3), dtype=np.uint8)
    assert h.rth_decode_image_rgb8(buf.ctypes.data, buf.size, C.byref(w), C.byref(hh), rgb.ctypes.data) == 0
    return rgb


_JPEG = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg.npz"))


@pytest.mark.parametrize("name", [str(n) for n in _JPEG["names"]])
def test_jpeg_texels_match_reference_decoder(name):
    """JPEG textures decode to the very bytes the reference's decoder yields (fixtures: tests/golden/make_jpeg_golden.py):
    baseline and progressive, every chroma layout, restart intervals, grey / CMYK / YCCK / RGB-tagged files."""
    want = _JPEG["rgb_" + name]
    got = _decode_image(_JPEG["file_" + name])
    assert got is not None, R.host_lib().rth_last_error()
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} bytes differ, max {np.abs(got.astype(int) - want).max()}"


def test_jpeg_rejects_garbage():
    assert _decode_image(b"\xff\xd8\xff\xd9") is None          # no frame
    assert _decode_image(b"not an image at all") is None
    ok = bytes(_JPEG["file_pil_444"])
    assert _decode_image(ok[:40]) is None                       # cut before the frame header
    sof = ok.index(b"\xff\xc0")
    assert _decode_image(ok[:sof + 4] + b"\x0c" + ok[sof + 5:]) is None   # 12-bit samples


_PNG = np.load(os.path.join(os.path.dirname(__file__), "golden", "png.npz"))


@pytest.mark.parametrize("name", [str(n) for n in _PNG["names"]])
def test_png_texels_match_reference_decoder(name):
    """PNG textures of every colour type, bit depth, row filter and interlace method decode to the bytes the reference's
    decoder yields (fixtures: tests/golden/make_png_golden.py)."""
    want = _PNG["rgb_" + name]
    got = _decode_image(_PNG["file_" + name])
    assert got is not None, R.host_lib().rth_last_error()
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} bytes differ"


def test_write_outputs_follows_the_reference(tmp_path):
    """One EXR for an .exr camera plus one PNG per tone map (src/main.cpp:561-607); a plain PNG otherwise."""
    import raytracer_amd as R
    from raytracer_amd import scenegen
    from oracle_lib import Oracle
    sc = R.Scene(scenegen.write_path_traced_scene(str(tmp_path / "s"), n=24, width=48, height=27, spp=4))
    img, _ = Oracle(sc).render(sc.camera(0), threads=2)
    paths = R.write_outputs(sc, 0, img, out_dir=str(tmp_path / "o"))
    assert [os.path.basename(p) for p in paths] == ["pt1m.exr", "pt1m_phot.png"]
    assert np.array_equal(png_pixels(paths[1]), sc.tonemap(0, 0, img)[::-1])
    sc2 = R.Scene(os.path.join(ROOT, "scenes", "simple.json"))
    img2, _ = Oracle(sc2).render(sc2.camera(0), threads=2)
    assert [os.path.basename(p) for p in R.write_outputs(sc2, 0, img2, out_dir=str(tmp_path / "o"))] == ["simple.png"]


@pytest.mark.parametrize("tree", ["from_the_host", "none"])
def test_shadow_ray_order_decision_on_the_host(tree, tmp_path, monkeypatch):
    """rt_shadow_order_free (host only): shadow rays may be walked in any order exactly when no surface can lie within the
    shadow-ray epsilon of a light sample (such a hit ends a shadow ray without counting as shadow, src/ray_trace.cpp:232-241,
    so only then does the order of the walk matter).  Lights that hang free of the geometry: yes; emitters made of triangles
    and a point light moved onto a vertex of a mesh: no.  A scene that comes without a tree (RT_GPU_BVH=1: one is built on
    the device) is judged by its primitives' boxes, with the same answers."""
    import ctypes as C
    import json
    import os
    import raytracer_amd as R
    from raytracer_amd import scenegen
    from golden import scenes as GS
    if tree == "none":
        monkeypatch.setenv("RT_GPU_BVH", "1")
    g = R.gpu_lib()
    g.rt_shadow_order_free.restype = C.c_int32
    g.rt_shadow_order_free.argtypes = [C.c_void_p]
    def free(path):
        sc = R.Scene(path)
        assert (sc.desc.n_bvh_nodes == 0) == (tree == "none")
        return bool(g.rt_shadow_order_free(C.byref(sc.desc)))
    out = str(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert free(os.path.join(root, "scenes", "simple.json"))
    assert free(GS.SCENES["brdfs"][0](out)) and free(GS.SCENES["distributed"][0](out)) and free(GS.SCENES["envmap"][0](out))
    assert not free(GS.SCENES["emitters"][0](out)) and not free(GS.SCENES["pathtrace"][0](out))
    assert not free(GS.SCENES["pt_uniform"][0](out))   # its sphere light's samples aim at the world's origin, on the ground mesh
    assert free(scenegen.write_path_traced_scene(out + "/pt", n=32, width=32, height=18, spp=4))
    assert free(scenegen.write_textured_scene(out + "/tex", n=16, width=32, height=18, spp=4))
    doc = json.load(open(os.path.join(root, "scenes", "simple.json")))
    verts = [float(x) for x in doc["Scene"]["VertexData"]["_data"].split()]
    doc["Scene"]["Lights"]["PointLight"]["Position"] = " ".join(f"{v:.6f}" for v in verts[0:3])
    path = str(tmp_path / "light_on_wall.json")
    json.dump(doc, open(path, "w"))
    assert not free(path)
