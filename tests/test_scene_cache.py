"""The binary scene cache (RT_SCENE_CACHE=<directory>): a scene loaded from its cache file is, array for array and
byte for byte, the scene a fresh load builds; a change to the scene file or to any file it pulls in (PLY mesh,
texture) makes the loader build it afresh; a damaged cache file is ignored."""
import ctypes as C
import os
import time

import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import _abi as A, scenegen
from golden import scenes as GS

ARRAYS = [("triangles", "n_triangles", A.Triangle), ("instances", "n_instances", A.Instance), ("spheres", "n_spheres", A.Sphere),
          ("planes", "n_planes", A.Plane), ("materials", "n_materials", A.Material), ("textures", "n_textures", A.Texture),
          ("texture_refs", "n_texture_refs", C.c_int32), ("images", "n_images", A.Image), ("texels_u8", "n_texels_u8", C.c_uint8),
          ("texels_f32", "n_texels_f32", C.c_float), ("lights", "n_lights", A.Light), ("bvh_nodes", "n_bvh_nodes", A.BvhNode),
          ("bvh_prims", "n_bvh_prims", A.BvhPrim)]
SCALARS = ["intersection_epsilon", "shadow_ray_epsilon", "max_depth", "background_texture", "is_probe"]


def snapshot(sc):
    """Everything rt_scene_create and the output stage get to see of a loaded scene, as bytes."""
    d = sc.desc
    out = {}
    for field, count, rec in ARRAYS:
        n = getattr(d, count) * C.sizeof(rec)
        out[field] = C.string_at(C.cast(getattr(d, field), C.c_void_p), n) if n else b""
    for k in SCALARS:
        out[k] = getattr(d, k)
    out["ambient"] = bytes(d.ambient_light)
    out["background"] = bytes(d.background_color)
    out["cameras"] = [(bytes(sc.camera(i)), sc.image_name(i), [sc.tonemap_extension(i, t) for t in range(sc.tonemap_count(i))])
                      for i in range(sc.n_cameras)]
    return out


def cache_files(d):
    return sorted(f for f in os.listdir(d) if f.endswith(".rtscene")) if os.path.isdir(d) else []


@pytest.mark.parametrize("name", ["textures", "transforms", "pathtrace", "env_probe", "textured"])
def test_cached_scene_equals_fresh_scene(name, tmp_path, monkeypatch):
    path = GS.SCENES[name][0](str(tmp_path / "scene"))
    monkeypatch.delenv("RT_SCENE_CACHE", raising=False)
    fresh = snapshot(R.Scene(path))
    monkeypatch.setenv("RT_SCENE_CACHE", str(tmp_path / "cache"))
    first = R.Scene(path)                                          # builds and stores
    assert len(cache_files(tmp_path / "cache")) == 1 and first.timings()["bvh_ms"] > 0 or first.desc.n_bvh_nodes <= 1
    cached = R.Scene(path)                                         # mapped
    assert cached.timings()["bvh_ms"] == 0
    assert snapshot(first) == fresh and snapshot(cached) == fresh


def test_cache_follows_the_files_a_scene_pulls_in(tmp_path, monkeypatch):
    monkeypatch.setenv("RT_SCENE_CACHE", str(tmp_path / "cache"))
    path = GS.SCENES["textured"][0](str(tmp_path / "scene"))
    R.Scene(path)
    assert R.Scene(path).timings()["bvh_ms"] == 0                  # served from the cache
    before = snapshot(R.Scene(path))
    # a texture changes (same size, same name): the cache file no longer matches and is replaced
    GS.write_png(str(tmp_path / "scene" / "marble.png"), 255 - scenegen._marble_image())
    again = R.Scene(path)
    assert again.timings()["bvh_ms"] > 0 and snapshot(again)["texels_u8"] != before["texels_u8"]
    assert R.Scene(path).timings()["bvh_ms"] == 0 and len(cache_files(tmp_path / "cache")) == 1
    # the mesh changes
    scenegen.write_displaced_sphere_ply(str(tmp_path / "scene" / "m24_outward.ply"), 23, outward=True)
    changed = R.Scene(path)
    assert changed.timings()["bvh_ms"] > 0 and changed.desc.n_triangles == 2 * 23 * 23 + 4 != again.desc.n_triangles
    # the scene file changes: another cache file
    text = open(path).read().replace('"NumSamples": "64"', '"NumSamples": "16"')
    open(path, "w").write(text)
    other = R.Scene(path)
    assert other.timings()["bvh_ms"] > 0 and other.camera(0).num_samples == 16 and len(cache_files(tmp_path / "cache")) == 2


def test_damaged_cache_file_is_ignored(tmp_path, monkeypatch):
    monkeypatch.setenv("RT_SCENE_CACHE", str(tmp_path / "cache"))
    path = GS.SCENES["transforms"][0](str(tmp_path / "scene"))
    fresh = snapshot(R.Scene(path))
    (name,) = cache_files(tmp_path / "cache")
    full = open(tmp_path / "cache" / name, "rb").read()
    for damage in (full[:100], full[:len(full) // 2], b"\0" * len(full), full[:4] + b"\xff" * 60 + full[64:]):
        open(tmp_path / "cache" / name, "wb").write(damage)
        sc = R.Scene(path)
        assert sc.timings()["bvh_ms"] > 0 and snapshot(sc) == fresh   # built afresh (and the cache file rewritten)
        assert open(tmp_path / "cache" / name, "rb").read() == full


def test_cache_file_with_changed_bytes_is_ignored(tmp_path, monkeypatch):
    """The header carries a hash of everything behind it: one flipped bit anywhere in the arrays (a triangle, a tree node, a
    texel, the cameras), a byte appended, or a file written by a library built from other host sources (its stamp in the
    header) makes the load fall back to parsing, and the entry is rewritten."""
    monkeypatch.setenv("RT_SCENE_CACHE", str(tmp_path / "cache"))
    path = GS.SCENES["textures"][0](str(tmp_path / "scene"))
    fresh = snapshot(R.Scene(path))
    (name,) = cache_files(tmp_path / "cache")
    file = tmp_path / "cache" / name
    full = open(file, "rb").read()
    assert R.Scene(path).timings()["bvh_ms"] == 0          # served from the cache as it stands
    n = len(full)
    damaged = []
    for at in (n // 4, n // 2, 3 * n // 4, n - 1, 1024):
        b = bytearray(full)
        b[at] ^= 0x10
        damaged.append(bytes(b))
    damaged.append(full + b"\0")
    for d in damaged:
        open(file, "wb").write(d)
        sc = R.Scene(path)
        assert sc.timings()["bvh_ms"] > 0 and snapshot(sc) == fresh
        assert open(file, "rb").read() == full


def test_cached_load_of_the_million_triangle_scene_is_fast(tmp_path, monkeypatch):
    """What the cache is for: the benchmark scene (1 008 202 triangles, 1.7 M BVH nodes, 171 MB of arrays) loads from
    its cache file in well under 50 ms (about 7 ms when the file is in the page cache), against 0.7 s for a fresh load."""
    monkeypatch.setenv("RT_SCENE_CACHE", str(tmp_path / "cache"))
    path = scenegen.write_path_traced_scene(str(tmp_path / "scene"), width=64, height=36, spp=4)
    t0 = time.perf_counter()
    fresh = R.Scene(path)
    t_fresh = time.perf_counter() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        cached = R.Scene(path)
        best = min(best, time.perf_counter() - t0)
    assert cached.desc.n_triangles == 1008202 and cached.timings()["bvh_ms"] == 0
    assert best < 0.05 and best < t_fresh / 5
    assert snapshot(cached) == snapshot(fresh)
