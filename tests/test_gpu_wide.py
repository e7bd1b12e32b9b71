"""The trace kernel's two walks on the GPU: renders without the work counters walk the wide tree (four quantised boxes
per record, leaves filtered by their exact box), renders with the counters walk the two-box records, whose visits are
the reference's (src/bvh.cpp:383-480).  Both must find the same primitives at the same distances."""
import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import scenegen
from golden import scenes as GS

pytestmark = pytest.mark.gpu


def path_of(name, scene_dir):
    if name == "mesh70k":
        return scenegen.write_mesh_scene(scene_dir + "/mesh")
    if name == "pt1m":
        return scenegen.write_path_traced_scene(scene_dir + "/pt1m", width=320, height=180, spp=4)
    return GS.SCENES[name][0](scene_dir)


@pytest.mark.parametrize("name", list(GS.SCENES) + ["mesh70k", "pt1m"])
def test_wide_walk_equals_two_box_walk(name, scene_dir, monkeypatch):
    monkeypatch.setenv("RT_WIDE_NODES", "1")   # read at scene creation: the wide tree now, not after enough rays
    sc = R.Scene(path_of(name, scene_dir))
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    shape_w, id_w, t_w, _ = ren.trace_primary(cam)
    img_w, st_w = ren.render(cam)
    ren.set_counters(True)
    shape_2, id_2, t_2, _ = ren.trace_primary(cam)
    img_2, st_2 = ren.render(cam)
    assert st_2.nodes_visited > 0 and st_w.nodes_visited == 0          # which walk ran
    assert np.array_equal(shape_w, shape_2) and np.array_equal(id_w, id_2)
    assert np.array_equal(t_w.view(np.uint32), t_2.view(np.uint32))
    assert (st_w.primary_rays, st_w.secondary_rays) == (st_2.primary_rays, st_2.secondary_rays)
    # images: the same paths, summed by float atomics in a different order
    err = np.abs(img_w - img_2) / np.maximum(np.abs(img_2), 1 / 255)
    assert err.max() <= 1e-5


def test_the_switch_to_the_wide_tree_is_invisible(scene_dir, monkeypatch):
    """By default a scene gets its wide tree once it has been asked for 32 camera paths per BVH node (the build is a few
    launches over the nodes, which a single small frame does not repay): the 70 K-triangle scene changes walks with its second
    1080p frame.  The frames before and after are the same."""
    monkeypatch.delenv("RT_WIDE_NODES", raising=False)
    sc = R.Scene(path_of("mesh70k", scene_dir))
    assert 1 * 1920 * 1080 < 32 * sc.desc.n_bvh_nodes < 2 * 1920 * 1080         # the second frame is the first on the wide tree
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    assert not ren.wide_records()
    frames = [ren.render(cam)[0]]
    assert not ren.wide_records()
    frames += [ren.render(cam)[0] for _ in range(3)]
    assert ren.wide_records()
    for f in frames[1:]:
        assert np.abs(f - frames[0]).max() <= 1e-5 * max(1.0, float(frames[0].max()))
    ids = [ren.trace_primary(cam) for _ in range(2)]
    ren.set_counters(True)
    shape_2, id_2, t_2, _ = ren.trace_primary(cam)
    assert np.array_equal(ids[1][0], shape_2) and np.array_equal(ids[1][2].view(np.uint32), t_2.view(np.uint32))


def test_device_builder_declines_what_the_walk_cannot_hold():
    """The builder of the four-box records on the device against the same hand-made trees as the host's builder
    (tests/test_wide_tree.py): a chain of 40 links gets 20 records; one of 400 would leave 200 entries waiting on a stack of
    64 and is declined, as is a tree with a box that is not finite (such scenes render from the two-box records)."""
    import ctypes as C
    from test_wide_tree import chain_tree
    g = R.gpu_lib()
    g.rt_selftest_wide_records.restype = C.c_int64
    g.rt_selftest_wide_records.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    nodes, n = chain_tree(40)
    assert g.rt_selftest_wide_records(nodes, n, 64) == 20
    nodes, n = chain_tree(400)
    assert g.rt_selftest_wide_records(nodes, n, 64) == -1
    assert g.rt_selftest_wide_records(nodes, n, 1000) == 200
    nodes, n = chain_tree(40, bad_box=True)
    assert g.rt_selftest_wide_records(nodes, n, 64) == -1
    nodes, n = chain_tree(41)                                   # an odd chain: the last record holds two leaves
    assert g.rt_selftest_wide_records(nodes, n, 64) == 21


@pytest.mark.parametrize("name", ["materials", "transforms", "mesh70k"])
def test_rays_parallel_to_an_axis(name, scene_dir, monkeypatch):
    """An odd image width puts a column of pixels on the camera's vertical plane: those rays have a direction component
    that is exactly zero, so one reciprocal is infinite and slab products are +-inf (or NaN on a box face).  Both GPU
    walks and the CPU oracle must agree on them too."""
    from oracle_lib import Oracle
    monkeypatch.setenv("RT_WIDE_NODES", "1")
    sc = R.Scene(path_of(name, scene_dir))
    cam = sc.camera(0, resx=97, resy=65)
    ren = R.Renderer(sc)
    shape_w, id_w, t_w, _ = ren.trace_primary(cam)
    ren.set_counters(True)
    shape_2, id_2, t_2, _ = ren.trace_primary(cam)
    shape_o, id_o, t_o, _ = Oracle(sc).trace_primary(cam)
    assert np.array_equal(shape_w, shape_2) and np.array_equal(t_w.view(np.uint32), t_2.view(np.uint32))
    assert np.array_equal(shape_w, shape_o) and np.array_equal(id_w, id_o) and np.array_equal(t_w.view(np.uint32), t_o.view(np.uint32))
    assert (shape_w.reshape(65, 97)[:, 48] >= 0).any()       # the centre column sees the scene


def test_full_size_frame_same_on_both_walks(scene_dir, monkeypatch):
    """The benchmark scene at its full resolution (3840x2160, 1M triangles; 16 of its 256 samples per pixel): the wide
    walk's image against the two-box walk's, whose visits are the reference's."""
    monkeypatch.setenv("RT_WIDE_NODES", "1")
    sc = R.Scene(scenegen.write_path_traced_scene(scene_dir + "/pt1m_4k", width=3840, height=2160, spp=16))
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    assert ren.shadow_order_free()   # (so the first render's shadow rays walk in record order, the second's in the reference's)
    img_w, st_w = ren.render(cam)
    ren.set_counters(True)
    img_2, st_2 = ren.render(cam)
    assert st_w.nodes_visited == 0 and st_2.nodes_visited > 30 * st_2.total_rays // 2
    assert (st_w.primary_rays, st_w.secondary_rays) == (st_2.primary_rays, st_2.secondary_rays) and st_w.primary_rays == 3840 * 2160 * 16
    err = np.abs(img_w - img_2) / np.maximum(np.abs(img_2), 1 / 255)
    assert err.max() <= 1e-5


def test_shadow_rays_in_any_order_only_where_the_order_cannot_matter(scene_dir, repo_root, monkeypatch, tmp_path):
    """A shadow ray ends with the first hit before the light sample and counts as shadow when that hit lies more than the
    shadow-ray epsilon before it (src/ray_trace.cpp:232-241): the order of the walk matters exactly when a surface can lie
    within epsilon of a sample.  Scenes whose lights hang free of the geometry walk their shadow rays in record order;
    emitters made of triangles (their samples sit on them) and a point light on a wall keep the reference's order."""
    import json
    import os
    monkeypatch.setenv("RT_WIDE_NODES", "1")
    from test_oracle_vs_reference import scene_path
    # (pt_uniform: its sphere light is placed by its centre, not by a transformation, so its samples aim at model * origin =
    # the world's origin (src/lights.cpp:418-422), which lies on the ground mesh)
    free = {"simple": True, "brdfs": True, "distributed": True, "pt_uniform": False, "emitters": False, "pathtrace": False}
    for name, want in free.items():
        ren = R.Renderer(R.Scene(scene_path(name, scene_dir, repo_root)))
        assert ren.shadow_order_free() == want, name
    small = R.Scene(scenegen.write_path_traced_scene(scene_dir + "/pt_order", n=48, width=96, height=54, spp=16))
    assert R.Renderer(small).shadow_order_free()          # a sphere emitter: every sample aims at its centre, deep inside it
    textured = R.Scene(scenegen.write_textured_scene(scene_dir + "/tex_order", n=24, width=96, height=54, spp=4))
    assert R.Renderer(textured).shadow_order_free()       # area lights above the scene
    doc = json.load(open(os.path.join(repo_root, "scenes", "simple.json")))
    verts = [float(x) for x in doc["Scene"]["VertexData"]["_data"].split()] if isinstance(doc["Scene"]["VertexData"], dict) else [float(x) for x in doc["Scene"]["VertexData"].split()]
    lights = doc["Scene"]["Lights"]["PointLight"]
    first = lights[0] if isinstance(lights, list) else lights
    first["Position"] = " ".join(f"{v:.6f}" for v in verts[0:3])   # on a vertex of the mesh
    path = str(tmp_path / "light_on_wall.json")
    json.dump(doc, open(path, "w"))
    assert not R.Renderer(R.Scene(path)).shadow_order_free()


@pytest.mark.parametrize("name", ["simple", "mesh70k", "one_sphere_emitter", "textured_generated"] + sorted(GS.SCENES))
def test_shadow_rays_in_any_order_give_the_same_image(name, scene_dir, repo_root, monkeypatch):
    """Every scene whose shadow rays are walked in record order, against the same scene with the reference's order
    forced: the same image (bit for bit where no path splits; deterministic scenes and stochastic ones alike: same rays,
    same answers)."""
    monkeypatch.setenv("RT_WIDE_NODES", "1")
    from test_oracle_vs_reference import scene_path
    if name == "one_sphere_emitter":
        path = scenegen.write_path_traced_scene(scene_dir + "/pt_order", n=48, width=96, height=54, spp=16)
    elif name == "textured_generated":
        path = scenegen.write_textured_scene(scene_dir + "/tex_order", n=24, width=96, height=54, spp=4)
    else:
        path = scene_path(name, scene_dir, repo_root)
    if name in ("emitters", "pathtrace", "pt_uniform", "cornell", "meshlight"):
        pytest.skip("a surface lies within the shadow-ray epsilon of this scene's light samples: it keeps the reference's order")
    monkeypatch.setenv("RT_PAIR_GATHER", "0")
    monkeypatch.setenv("RT_ANY_ORDERED", "1")
    sc = R.Scene(path)
    ordered = R.Renderer(sc)
    assert not ordered.shadow_order_free()
    img0, st0 = ordered.render(sc.camera(0))
    monkeypatch.setenv("RT_ANY_ORDERED", "0")
    free = R.Renderer(sc)
    assert free.shadow_order_free()
    img1, st1 = free.render(sc.camera(0))
    # a flipped answer moves a pixel by a light sample's contribution; the order of a sample's float additions (paths that
    # split at a dielectric add to one sample from two lanes, in arrival order) by an ulp or two
    err = np.abs(img0 - img1) / np.maximum(np.abs(img0), 1.0 / 255.0)
    assert err.max() <= 2e-6, "a shadow ray's answer changed with the order of its walk"
    if name in ("simple", "mesh70k", "brdfs", "distributed", "textured_generated", "one_sphere_emitter"):   # no path splits
        assert np.array_equal(img0, img1)
    assert (st0.shadow_rays, st0.shadow_rays_traced, st0.secondary_rays) == (st1.shadow_rays, st1.shadow_rays_traced, st1.secondary_rays)
