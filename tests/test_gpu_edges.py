"""GPU edge cases through the C ABI: empty and ragged inputs, non-square sample counts, tiny queues that force
several batches and the overflow retry, and rejected descriptions.  Results are checked against the CPU oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import _abi as A
from golden import scenes as GS
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def patched(src, dst, edit):
    doc = json.load(open(src))
    edit(doc["Scene"])
    with open(dst, "w") as f:
        json.dump(doc, f)
    return dst


def same_as_oracle(path, tol=1e-5, **cam_overrides):
    sc = R.Scene(path)
    cam = sc.camera(0, **cam_overrides)
    ren = R.Renderer(sc)
    ren.set_counters(True)
    gpu, sg = ren.render(cam)
    cpu, so = Oracle(sc).render(cam)
    assert (sg.primary_rays, sg.shadow_rays, sg.secondary_rays) == (so.primary_rays, so.shadow_rays, so.secondary_rays)
    assert (sg.nodes_visited, sg.triangle_tests, sg.sphere_tests) == (so.nodes_visited, so.triangle_tests, so.sphere_tests)
    err = np.abs(gpu - cpu) / np.maximum(np.abs(cpu), 1 / 255)
    assert err.max() <= tol, err.max()   # every pixel
    return gpu, sg


def test_scene_without_objects_is_background(tmp_path):
    def edit(s):
        s["Objects"] = {}
        s["BackgroundColor"] = "10 20 30"
    path = patched(os.path.join(ROOT, "scenes", "simple.json"), str(tmp_path / "empty.json"), edit)
    img, st = same_as_oracle(path)
    assert st.shadow_rays == 0 and st.secondary_rays == 0 and st.nodes_visited == 0
    assert np.unique(img.reshape(-1, 3), axis=0).shape[0] == 1


def test_scene_without_lights(tmp_path):
    def edit(s):
        s["Lights"] = {"AmbientLight": "25 25 25"}
    same_as_oracle(patched(os.path.join(ROOT, "scenes", "simple.json"), str(tmp_path / "nolights.json"), edit))


@pytest.mark.parametrize("res", [(1, 1), (33, 31), (31, 65), (100, 7)])
def test_ragged_resolutions(res):
    """Frames that do not fill their last 32x32 tiles, down to a single pixel."""
    img, st = same_as_oracle(os.path.join(ROOT, "scenes", "simple.json"), resx=res[0], resy=res[1])
    assert img.shape == (res[1], res[0], 3) and st.primary_rays == res[0] * res[1]


def test_sample_count_that_is_not_a_square(scene_dir):
    """NumSamples 10 renders floor(sqrt(10))^2 = 9 samples (src/camera.cpp:149)."""
    path = GS.SCENES["distributed"][0](scene_dir)
    sc = R.Scene(path)
    cam = sc.camera(0, num_samples=10)
    img, st = same_as_oracle(path, num_samples=10)
    assert st.primary_rays == cam.resx * cam.resy * 9


@pytest.mark.parametrize("name", ["materials", "pathtrace", "distributed"])
def test_small_queues_force_batches_and_retries(name, scene_dir, monkeypatch):
    """With a small wavefront (256 Ki entries; one tile of a 64-sample camera needs 64 Ki, times its branching) a
    frame needs several batches, and scenes whose paths branch more than planned are retried.  The image must not
    depend on any of that."""
    path = GS.SCENES[name][0](scene_dir)
    sc = R.Scene(path)
    cam = sc.camera(0)
    big, sb = R.Renderer(sc).render(cam)
    monkeypatch.setenv("RT_WAVEFRONT_CAPACITY", "262144")
    small, ss = R.Renderer(sc).render(cam)
    assert ss.total_rays == sb.total_rays
    err = np.abs(small - big) / np.maximum(np.abs(big), 1 / 255)
    assert err.max() <= 1e-5


def test_leaf_with_two_hundred_entries(tmp_path):
    """200 coincident triangles have one centroid, so the builder cannot split them (src/bvh.cpp:150-300) and they share
    one leaf; the reference's leaf count is 16 bits wide (src/bvh.h:20).  Both walks against the oracle."""
    def edit(s):
        s["Objects"]["Mesh"]["Faces"]["_data"] = "3 1 2  1 3 4 " + " 5 6 7" * 200
        del s["Objects"]["Triangle"]
    path = patched(os.path.join(ROOT, "scenes", "simple.json"), str(tmp_path / "pile.json"), edit)
    sc = R.Scene(path)
    _, meta, prims = sc.bvh_arrays()
    assert meta[:, 1].max() >= 200                                  # (offset, n_prims, axis): one leaf holds the pile
    first_of_pile = prims[meta[np.argmax(meta[:, 1]), 0], 0]
    img, st = same_as_oracle(path)                                  # the counted walk, visit for visit
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    fast, sf = ren.render(cam)                                      # the timed path
    assert sf.nodes_visited == 0 and np.abs(fast - img).max() <= 1e-5 * max(1.0, float(img.max()))
    shape, ids, t, _ = ren.trace_primary(cam)
    s0, i0, t0, _ = Oracle(sc).trace_primary(cam)
    assert np.array_equal(shape, s0) and np.array_equal(t.view(np.uint32), t0.view(np.uint32))
    pile = (shape >= 2) & (shape < 202)                              # (shapes 0, 1: the quad; 202: the sphere)
    assert pile.sum() >= 4 and set(shape[pile]) == {first_of_pile}   # of equal hits the leaf's first entry is kept


def test_recursion_depth_beyond_sixty_four(scene_dir, tmp_path):
    """MaxRecursionDepth 100 (the reference takes any value, src/parser.cpp:1253-1258): mirrors and glass against the oracle."""
    def edit(s):
        s["MaxRecursionDepth"] = "100"
    path = patched(GS.SCENES["materials"][0](scene_dir), str(tmp_path / "deep.json"), edit)
    _, st = same_as_oracle(path)
    assert st.secondary_rays > 0


def test_scene_from_the_cache_renders_the_same(scene_dir, tmp_path, monkeypatch):
    """A scene mapped from its cache file (RT_SCENE_CACHE) goes to the device straight from the mapping."""
    path = GS.SCENES["textured"][0](scene_dir)
    monkeypatch.delenv("RT_SCENE_CACHE", raising=False)
    sc = R.Scene(path)
    fresh, _ = R.Renderer(sc).render(sc.camera(0))
    monkeypatch.setenv("RT_SCENE_CACHE", str(tmp_path / "cache"))
    R.Scene(path)
    cached_scene = R.Scene(path)
    assert cached_scene.timings()["bvh_ms"] == 0
    cached, _ = R.Renderer(cached_scene).render(cached_scene.camera(0))
    assert np.abs(cached - fresh).max() <= 1e-5 * max(1.0, float(fresh.max()))


def test_rejected_descriptions():
    g = R.gpu_lib()
    sc = R.Scene(os.path.join(ROOT, "scenes", "simple.json"))
    ptr = C.c_void_p()
    # wrong ABI version
    desc = A.SceneDesc.from_buffer_copy(sc.desc)
    desc.abi_version = 1
    assert g.rt_scene_create(C.byref(desc), C.byref(ptr)) != 0 and b"ABI" in g.rt_last_error()
    # a BVH node that points outside the node array
    desc = A.SceneDesc.from_buffer_copy(sc.desc)
    nodes = (A.BvhNode * desc.n_bvh_nodes)()
    C.memmove(nodes, desc.bvh_nodes, C.sizeof(nodes))
    interior = [i for i in range(desc.n_bvh_nodes) if nodes[i].n_prims == 0]
    assert interior
    nodes[interior[0]].offset = desc.n_bvh_nodes + 5
    desc.bvh_nodes = C.cast(nodes, C.POINTER(A.BvhNode))
    assert g.rt_scene_create(C.byref(desc), C.byref(ptr)) != 0 and b"BVH" in g.rt_last_error()
    # a leaf entry that names a triangle that does not exist
    desc = A.SceneDesc.from_buffer_copy(sc.desc)
    prims = (A.BvhPrim * desc.n_bvh_prims)()
    C.memmove(prims, desc.bvh_prims, C.sizeof(prims))
    prims[0].shape = desc.n_triangles + desc.n_spheres
    desc.bvh_prims = C.cast(prims, C.POINTER(A.BvhPrim))
    assert g.rt_scene_create(C.byref(desc), C.byref(ptr)) != 0
    # camera and argument checks
    ren = R.Renderer(sc)
    with pytest.raises(R.RaytracerError):
        ren.render(sc.camera(0, resx=0), out=np.zeros((1, 1, 3), dtype=np.float32))
    with pytest.raises(R.RaytracerError):
        ren.render(sc.camera(0), out=np.zeros((64, 64, 3), dtype=np.float32), rank=2, world=2)
    assert g.rt_render(None, None, None, None) != 0


def test_command_line_end_to_end(tmp_path, scene_dir):
    """./raytracer on a PNG scene and on an EXR scene with tone maps: the files the reference would write, with the
    pixels of a render through the ABI."""
    import ctypes
    import subprocess
    from test_host import png_pixels
    exe = os.path.join(ROOT, "raytracer")
    simple = os.path.join(ROOT, "scenes", "simple.json")
    r = subprocess.run([exe, simple], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "simple.png rendering done:" in r.stdout
    sc = R.Scene(simple)
    img, _ = R.Renderer(sc).render(sc.camera(0))
    assert np.array_equal(png_pixels(str(tmp_path / "outputs" / "simple.png")), R.gpu_quantize_ldr(img)[::-1])

    def edit(s):
        s["Cameras"]["Camera"]["Tonemap"] = [
            {"TMO": "Photographic", "TMOOptions": "0.18 2", "Saturation": "1.0", "Gamma": "2.2", "Extension": "_phot.png"},
            {"TMO": "ACES", "TMOOptions": "0.5 0", "Saturation": "1.0", "Gamma": "1.8", "Extension": "_aces.png"}]
    env = patched(GS.SCENES["envmap"][0](scene_dir), os.path.join(scene_dir, "envmap_tm.json"), edit)
    r = subprocess.run([exe, env, "0", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Saving tone mapped image: envmap_phot.png" in r.stdout
    h = R.host_lib()
    w_, h_ = ctypes.c_int32(), ctypes.c_int32()
    exr = os.fsencode(str(tmp_path / "outputs" / "envmap.exr"))
    assert h.rth_read_exr(exr, ctypes.byref(w_), ctypes.byref(h_), None) == 0
    rgba = np.empty((h_.value, w_.value, 4), dtype=np.float32)
    assert h.rth_read_exr(exr, ctypes.byref(w_), ctypes.byref(h_), rgba.ctypes.data) == 0
    frame = np.ascontiguousarray(rgba[:, :, :3])
    assert np.isfinite(frame).all() and frame.max() > 0
    assert np.array_equal(png_pixels(str(tmp_path / "outputs" / "envmap_phot.png")), R.gpu_tonemap(frame, 0, 0.18, 2.0, 1.0, 2.2)[::-1])
    assert np.array_equal(png_pixels(str(tmp_path / "outputs" / "envmap_aces.png")), R.gpu_tonemap(frame, 2, 0.5, 0.0, 1.0, 1.8)[::-1])


@pytest.mark.parametrize("env_extra,n", [({}, 12), ({"RT_GPU": "0,0,0"}, 12), ({"RT_GPU": "0,0,0"}, 13),
                                         ({"RT_GPUS_DEVICES": "0,0,0,0", "RT_GPUS": "2"}, 13),
                                         ({"RT_GPUS_DEVICES": "0,0,0,0", "RT_GPUS": "2", "RT_TAIL_TILES": "0"}, 13),
                                         ({"RT_GPU": "0,0,0,0,0,0,0,0"}, 3)])
def test_command_line_directory_of_frames(tmp_path, env_extra, n):
    """A directory of scene files (an animation: one JSON per frame) goes through the loader / render / writer
    pipeline; every frame's image is written and equals a render through the ABI, and each file's log block stays
    in one piece, the blocks in file order.  Rehearsals of a node with several GPUs on this box's one (a device named
    several times):
      RT_GPU=0,0,0                      three render threads take the frames in turn; a thirteenth frame is left over
                                        and rendered by all three as a team (split by tiles, gathered)
      RT_GPUS_DEVICES=0,0,0,0 RT_GPUS=2 two teams of two beside each other: frame-and-tile parallel; the frame left over
                                        goes to one team of four (RT_TAIL_TILES=0: to a team of two, the other idles)
      eight devices, three frames       fewer frames than GPUs: three teams of 3 + 3 + 2, every frame split by tiles"""
    import subprocess
    from test_host import png_pixels
    frames = tmp_path / "frames"
    frames.mkdir()
    for k in range(n):
        def edit(s, k=k):
            cam = s["Cameras"]["Camera"]
            cam["Position"] = f"{0.05 * k} 0 0"
            cam["ImageName"] = f"anim_{k:04d}.png"
        patched(os.path.join(ROOT, "scenes", "simple.json"), str(frames / f"anim_{k:04d}.json"), edit)
    env = dict(os.environ, **env_extra)
    r = subprocess.run([os.path.join(ROOT, "raytracer"), str(frames)], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert sum("rendering done:" in l for l in lines) == n
    starts = [i for i, l in enumerate(lines) if l.startswith("JSON parsing started")]
    assert len(starts) == n
    done = []
    for a, b in zip(starts, starts[1:] + [len(lines)]):            # one "started" and one "done" per block
        block = lines[a:b]
        assert sum(l.endswith(".png started") for l in block) == 1 and sum("rendering done:" in l for l in block) == 1
        done += [l.split()[0] for l in block if "rendering done:" in l]
    assert sorted(done) == [f"anim_{k:04d}.png" for k in range(n)]
    for k in (0, n // 2, n - 1) if not env_extra else range(n):
        sc = R.Scene(str(frames / f"anim_{k:04d}.json"))
        img, _ = R.Renderer(sc).render(sc.camera(0))
        assert np.array_equal(png_pixels(str(tmp_path / "outputs" / f"anim_{k:04d}.png")), R.gpu_quantize_ldr(img)[::-1])


def test_command_line_team_reports_a_failed_frame_and_exits(tmp_path):
    """A frame that a rank of the team cannot render (a camera without samples) ends the run with the reference's exit
    behaviour -- message, non-zero status -- instead of leaving the other ranks waiting for its tiles."""
    import subprocess

    def edit(s):
        s["Cameras"]["Camera"]["NumSamples"] = "0"
    path = patched(os.path.join(ROOT, "scenes", "simple.json"), str(tmp_path / "bad.json"), edit)
    r = subprocess.run([os.path.join(ROOT, "raytracer"), path], cwd=tmp_path, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, RT_GPUS_DEVICES="0,0,0"))
    assert r.returncode != 0 and "rt_render_gather failed" in r.stdout, r.stdout + r.stderr


def test_repeated_frame_is_replayed_from_a_graph_with_the_same_result(tmp_path):
    """A frame of one batch that is asked for again with the same camera and buffers is captured into a HIP graph (second
    request) and replayed (third and later): every repetition gives the first frame's image and statistics, a different
    camera in between goes the ordinary way, and RT_FRAME_GRAPH=0 gives the same image."""
    import subprocess, sys
    sc = R.Scene(os.path.join(ROOT, "scenes", "simple.json"))
    ren = R.Renderer(sc)
    cam = sc.camera(0)
    first, st0 = ren.render(cam)
    for _ in range(4):
        img, st = ren.render(cam)
        assert np.array_equal(img, first)
        assert (st.primary_rays, st.shadow_rays, st.secondary_rays, st.shaded_hits) == (st0.primary_rays, st0.shadow_rays, st0.secondary_rays, st0.shaded_hits)
    other = sc.camera(0)
    other.position[0] += 0.25
    moved, _ = ren.render(other)
    assert not np.array_equal(moved, first)
    again, _ = ren.render(cam)
    assert np.array_equal(again, first)
    moved2, _ = ren.render(other)
    moved3, _ = ren.render(other)
    assert np.array_equal(moved2, moved) and np.array_equal(moved3, moved)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import raytracer_amd as R; sc = R.Scene(%r); ren = R.Renderer(sc); cam = sc.camera(0);"
            "imgs = [ren.render(cam)[0] for _ in range(3)]; np.save(%r, imgs[2]); assert all(np.array_equal(i, imgs[0]) for i in imgs)"
            % (ROOT, os.path.join(ROOT, "scenes", "simple.json"), str(tmp_path / "plain.npy")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RT_FRAME_GRAPH="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.load(tmp_path / "plain.npy"), first)


def test_a_sequence_of_scenes_holds_on_to_nothing(scene_dir, repo_root, monkeypatch):
    """One scene per file is how a directory of frames runs: create, render, destroy, a few hundred times.  The buffers of a
    destroyed scene go to a bounded cache (blocks, streams, a parked wavefront state) that the next scene draws on, so the
    free device memory settles after the first few scenes whatever their order, and every frame is the first one's."""
    from test_oracle_vs_reference import scene_path
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")   # (the runtime the library runs on is loaded already)
    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    scenes = [R.Scene(scene_path(name, scene_dir, repo_root)) for name in ("simple", "mesh70k", "materials")]
    monkeypatch.setenv("RT_GPU_BVH", "1")             # (the same scenes without a tree: theirs is built on the device, from the same caches)
    scenes += [R.Scene(scene_path(name, scene_dir, repo_root)) for name in ("simple", "mesh70k")]
    monkeypatch.delenv("RT_GPU_BVH")
    first = {}
    settled = None
    for turn in range(90):
        for mode in ("0", "1"):                       # without and with the four-box records
            sc = scenes[turn % 3] if turn % 2 == 0 or turn % 3 == 2 else scenes[3 + turn % 3]
            monkeypatch.setenv("RT_WIDE_NODES", mode)
            ren = R.Renderer(sc)
            img, _ = ren.render(sc.camera(0, resx=64, resy=48))
            ren.close()
            key = (turn % 3, mode, sc.desc.n_bvh_nodes == 0)
            if key in first:   # (a scene whose paths split adds a sample's parts in arrival order: equal to the order of float additions)
                if turn % 3 == 2:
                    assert (np.abs(img - first[key]) / np.maximum(np.abs(first[key]), 1 / 255)).max() <= 1e-5, key
                else:
                    assert np.array_equal(img, first[key]), key
            else:
                first[key] = img.copy()
        if turn == 20:
            settled = free_bytes()
    assert settled - free_bytes() < (64 << 20), "device memory keeps going with every scene"
