"""The tree built on the device (a scene loaded with RT_GPU_BVH=1 carries no tree; rt_scene_create builds one: Morton
order, clusters merged bottom-up, leaves of up to four primitives, two-box and four-box records) against the reference's
SAH tree built on the host, for the same scene.

The two trees hold the same primitives in different shapes, so rays meet them in a different order.  What must agree:
hit or miss for every primary ray; the hit distance; and the primitive, except where two primitives are hit at exactly
the same distance (the reference keeps whichever it meets first, src/bvh.cpp:445, and that depends on the tree).
Images agree except at such pixels and within the re-association of float sums."""
import os

import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import scenegen
from golden import scenes as GS
from oracle_lib import Oracle
from test_oracle_vs_reference import scene_path

pytestmark = pytest.mark.gpu


def two_scenes(path, monkeypatch):
    monkeypatch.delenv("RT_GPU_BVH", raising=False)
    host_tree = R.Scene(path)
    monkeypatch.setenv("RT_GPU_BVH", "1")
    no_tree = R.Scene(path)
    monkeypatch.delenv("RT_GPU_BVH")
    assert no_tree.desc.n_bvh_nodes == 0 and (no_tree.desc.n_bvh_prims == host_tree.desc.n_bvh_prims)
    return host_tree, no_tree


@pytest.mark.parametrize("walk", ["four_box", "two_box"])
@pytest.mark.parametrize("name", ["simple", "mesh70k"] + sorted(GS.SCENES))
def test_device_tree_finds_the_same_hits(name, walk, scene_dir, repo_root, monkeypatch):
    """Both sets of records the builder writes: the four-box records every uncounted frame walks, and (RT_WIDE_NODES=0, or
    with the work counters on) the two-box records."""
    if walk == "two_box":
        monkeypatch.setenv("RT_WIDE_NODES", "0")
    host_tree, no_tree = two_scenes(scene_path(name, scene_dir, repo_root), monkeypatch)
    cam = host_tree.camera(0)
    s0, i0, t0, _ = Oracle(host_tree).trace_primary(cam)              # the reference's tree, walked on the host
    ren = R.Renderer(no_tree)
    if walk == "two_box" or no_tree.desc.n_bvh_prims > 4:             # (a tree of up to four primitives may be a single leaf)
        assert ren.wide_records() == (walk == "four_box")
    s1, i1, t1, _ = ren.trace_primary(cam)
    assert np.array_equal(s0 >= 0, s1 >= 0), "hit / miss differs"
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)), "closest distances differ"
    other = (s0 != s1) | (i0 != i1)                                    # same distance, another primitive: ties only
    # (a 64x64 frame of two triangles that share a diagonal: every pixel on it is a tie)
    assert other.mean() <= (5e-3 if name == "simple" else 2e-3), f"{other.sum()} of {other.size} primary hits resolve to another primitive"
    ref, st0 = R.Renderer(host_tree).render(cam)
    img, st1 = ren.render(cam)
    assert st0.primary_rays == st1.primary_rays
    err = (np.abs(img - ref) / np.maximum(np.abs(ref), 1 / 255)).max(axis=2)
    if name in ("cornell", "meshlight"):
        # lit by emitter triangles alone: a shadow ray towards a sample on the emitter finds the emitter itself within the
        # shadow-ray epsilon of the sample, and whether that hit or a real occluder ends the walk depends on the order of the
        # walk (src/ray_trace.cpp:232-241) -- another tree, another order.  The frames agree in brightness.
        assert abs(img.mean() / ref.mean() - 1.0) <= 0.02, (img.mean(), ref.mean())
        return
    assert (err > 1e-5).mean() <= 5e-3, f"{(err > 1e-5).sum()} pixels differ (max {err.max()})"


def test_many_primitives_with_one_centre(tmp_path, monkeypatch):
    """200 coincident triangles share a Morton code: the builder tells them apart by position and still makes a shallow tree."""
    import json
    doc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "simple.json")))
    doc["Scene"]["Objects"]["Mesh"]["Faces"]["_data"] = "3 1 2  1 3 4 " + " 5 6 7" * 200
    del doc["Scene"]["Objects"]["Triangle"]
    path = str(tmp_path / "pile.json")
    json.dump(doc, open(path, "w"))
    host_tree, no_tree = two_scenes(path, monkeypatch)
    cam = host_tree.camera(0)
    s0, _, t0, _ = R.Renderer(host_tree).trace_primary(cam)
    s1, _, t1, _ = R.Renderer(no_tree).trace_primary(cam)
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    pile = (s0 >= 2) & (s0 < 202)
    assert pile.sum() >= 4 and ((s1 >= 2) & (s1 < 202))[pile].all()    # some triangle of the pile, at the same distance


def test_million_triangle_scene_on_the_device_tree(scene_dir, monkeypatch):
    """The benchmark scene: primary hits of the device-built tree against the host tree's at 960x540, and the build time."""
    import time
    path = scenegen.write_path_traced_scene(scene_dir + "/pt1m_gpu_bvh", width=960, height=540, spp=4)
    host_tree, no_tree = two_scenes(path, monkeypatch)
    cam = host_tree.camera(0)
    s0, i0, t0, _ = R.Renderer(host_tree).trace_primary(cam)
    t_begin = time.perf_counter()
    ren = R.Renderer(no_tree)
    build_s = time.perf_counter() - t_begin
    s1, i1, t1, _ = ren.trace_primary(cam)
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and ((s0 != s1) | (i0 != i1)).mean() <= 1e-4
    assert build_s < 1.0                                               # upload + build of 1 M primitives (measured: tens of ms)
    img, _ = ren.render(cam)
    ref, _ = R.Renderer(host_tree).render(cam)
    err = (np.abs(img - ref) / np.maximum(np.abs(ref), 1 / 255)).max(axis=2)
    assert (err > 1e-5).mean() <= 1e-3


@pytest.mark.parametrize("n_tris", [1, 2, 3, 5, 9])
def test_device_tree_of_a_handful_of_primitives(n_tris, tmp_path, monkeypatch):
    """One primitive (the root is a leaf), two or three (the whole tree collapses into one leaf), five and nine (the
    first real interior nodes): the device-built tree finds the host tree's hits."""
    import json
    rng = np.random.default_rng(n_tris)
    verts, faces = [], []
    for k in range(n_tris):
        c = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), -2.0 - 0.1 * k])
        for d in ((-0.3, -0.3, 0), (0.3, -0.3, 0), (0, 0.3, 0)):
            verts.append(c + np.array(d))
        faces.append(f"{3 * k + 1} {3 * k + 2} {3 * k + 3}")
    doc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "simple.json")))
    doc["Scene"]["VertexData"]["_data"] = "  ".join(f"{v[0]:.4f} {v[1]:.4f} {v[2]:.4f}" for v in verts)
    doc["Scene"]["Objects"] = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": "  ".join(faces)}}}
    path = str(tmp_path / "few.json")
    json.dump(doc, open(path, "w"))
    host_tree, no_tree = two_scenes(path, monkeypatch)
    cam = host_tree.camera(0)
    s0, i0, t0, _ = R.Renderer(host_tree).trace_primary(cam)
    ren = R.Renderer(no_tree)
    s1, i1, t1, _ = ren.trace_primary(cam)
    assert (s0 >= 0).sum() > 20
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(s0, s1)   # (distinct depths: no ties)
    img0, _ = R.Renderer(host_tree).render(cam)
    img1, _ = ren.render(cam)
    assert np.array_equal(img0, img1)
