"""The render path that is timed -- work counters off: wide-tree walk, identical shadow rays traced once, sphere
emitters evaluated after their shadow ray -- directly against the CPU oracle, scene by scene, and at the benchmark's
full resolution.  (tests/test_gpu_parity.py checks the counted mode, whose traversal work equals the reference's
visit for visit; this file checks what bench.py measures.)

Bars: primary-hit (shape, instance id) and distance bit-exact; primary and secondary ray counts equal; every pixel
value within 1e-5 relative (floor 1/255), no outliers allowed -- the device carries path throughput forward and adds
each contribution to the sample with a float atomic, the oracle sums on the way back up its recursion, so sums are
re-associated and nothing else differs (measured: at most 4e-7, tools/outlier_survey.py)."""
import os

import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import scenegen
from golden import scenes as GS
from oracle_lib import Oracle
from test_oracle_vs_reference import scene_path

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0 / 255.0)


@pytest.mark.parametrize("name", ["simple", "mesh70k"] + sorted(GS.SCENES))
def test_timed_path_matches_oracle(name, scene_dir, repo_root, monkeypatch):
    monkeypatch.setenv("RT_WIDE_NODES", "1")   # the wide tree from the first ray on, as in a benchmark frame
    sc = R.Scene(scene_path(name, scene_dir, repo_root))
    cam = sc.camera(0)
    orc = Oracle(sc)
    ren = R.Renderer(sc)                       # counters stay off
    assert ren.wide_records() == (sc.desc.n_bvh_nodes >= 3), "the scene did not get its four-box records"
    s0, i0, t0, _ = orc.trace_primary(cam)
    s1, i1, t1, st = ren.trace_primary(cam)
    assert st.nodes_visited == 0, "the counted walk ran"
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1)
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)), "hit distances are not bit-exact"
    img0, so = orc.render(cam)
    img1, sg = ren.render(cam)
    assert sg.nodes_visited == 0
    assert (so.primary_rays, so.secondary_rays) == (sg.primary_rays, sg.secondary_rays)
    assert sg.shadow_rays <= so.shadow_rays and sg.shadow_rays_traced <= so.shadow_rays
    err = rel_err(img1, img0).max(axis=2)
    assert err.max() <= 1e-5, f"{(err > 1e-5).sum()} pixels over 1e-5; max {err.max()}"


def test_full_size_primary_hits_match_oracle(scene_dir, monkeypatch):
    """The benchmark scene (1M triangles) at 3840x2160: all 8.3 M primary hits of the wide walk against the oracle."""
    monkeypatch.setenv("RT_WIDE_NODES", "1")
    sc = R.Scene(scenegen.write_path_traced_scene(scene_dir + "/pt1m_4k", width=3840, height=2160, spp=16))
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    assert ren.wide_records()
    s1, i1, t1, st = ren.trace_primary(cam)
    s0, i0, t0, _ = Oracle(sc).trace_primary(cam)
    assert st.nodes_visited == 0 and len(s1) == 3840 * 2160 and (s1 >= 0).sum() > 4_000_000
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1)
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))


def test_mesh_scene_known_answers_at_1080p(scene_dir, monkeypatch):
    """Known answers recorded from the compiled reference for the 70 K-triangle scene at 1920x1080: 1 212 233 primary
    hits in each of three differently optimised builds, pixel (960, 540) sees shape 51182 of mesh 1 at t = 2.29568005.
    The reference's builds disagree with one another on 1 to 10 silhouette pixels of this frame (fast-math, approximate
    reciprocal square root); the strict evaluation here differs from them by one hit."""
    monkeypatch.setenv("RT_WIDE_NODES", "1")
    sc = R.Scene(scenegen.write_mesh_scene(scene_dir + "/kat"))
    cam = sc.camera(0)
    assert (cam.resx, cam.resy) == (1920, 1080)
    shape, ids, t, _ = R.Renderer(sc).trace_primary(cam)
    assert abs(int((shape >= 0).sum()) - 1212233) <= 2
    px = 540 * 1920 + 960
    assert (shape[px], ids[px]) == (51182, 1) and abs(float(t[px]) - 2.29568005) < 1e-5
    s0, i0, t0, _ = Oracle(sc).trace_primary(cam)
    assert np.array_equal(s0, shape) and np.array_equal(i0, ids) and np.array_equal(t0.view(np.uint32), t.view(np.uint32))


@pytest.mark.parametrize("name", ["pathtrace", "pt_uniform", "pt_env", "emitters", "one_sphere_emitter"])
def test_gather_beside_shading_changes_nothing(name, scene_dir, repo_root, monkeypatch):
    """Path-traced frames run the gather of a generation on a second stream beside the next generation's shading
    (RT_PAIR_GATHER=0: everything in turn, 2: also for small batches).  Same rays, same counts; the image agrees to the order of a sample's float
    additions, and with the oracle to the bar of the timed path."""
    monkeypatch.setenv("RT_WIDE_NODES", "1")
    if name == "one_sphere_emitter":   # the benchmark scene in small: the shading kernel writes the emitter's shadow rays itself
        sc = R.Scene(scenegen.write_path_traced_scene(scene_dir + "/pt_small", n=48, width=96, height=54, spp=16))
    else:
        sc = R.Scene(scene_path(name, scene_dir, repo_root))
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    monkeypatch.setenv("RT_PAIR_GATHER", "0")
    img0, st0 = ren.render(cam)
    monkeypatch.setenv("RT_PAIR_GATHER", "2")   # (by itself only for batches of 4 Mi slots and more)
    img1, st1 = ren.render(cam)
    img2, st2 = ren.render(cam)
    for st in (st1, st2):
        assert (st.primary_rays, st.secondary_rays, st.shadow_rays, st.shadow_rays_traced) == (
            st0.primary_rays, st0.secondary_rays, st0.shadow_rays, st0.shadow_rays_traced)
    assert rel_err(img1, img0).max() < 1e-5 and rel_err(img2, img0).max() < 1e-5
    ref, _ = Oracle(sc).render(cam)
    err = rel_err(img1, ref).max(axis=2)
    assert err.max() <= 1e-5, err.max()


@pytest.mark.parametrize("name", ["brdfs", "pt_uniform", "pt_env", "emitters", "one_sphere_emitter", "mesh70k"])
def test_shading_kernel_without_textures_and_mirrors_is_the_same_shading(name, scene_dir, repo_root, monkeypatch):
    """A scene with regular materials only and no texture is shaded by a build of the shading kernel that leaves the code
    for the other materials and for textures out (RT_PLAIN_SHADING=0 at scene creation: the full kernel).  The
    statements that remain are the same, no path splits in such a scene, and so the image is the same bit for bit."""
    monkeypatch.setenv("RT_WIDE_NODES", "1")
    monkeypatch.setenv("RT_PAIR_GATHER", "0")     # (beside the gather, a sample's additions come in arrival order)
    if name == "one_sphere_emitter":
        path = scenegen.write_path_traced_scene(scene_dir + "/pt_small", n=48, width=96, height=54, spp=16)
    elif name == "mesh70k":
        path = scenegen.write_mesh_scene(scene_dir + "/mesh")
    else:
        path = scene_path(name, scene_dir, repo_root)
    sc = R.Scene(path)
    cam = sc.camera(0)
    plain = R.Renderer(sc)
    monkeypatch.setenv("RT_PLAIN_SHADING", "0")
    full = R.Renderer(sc)
    if not plain.plain_shading():
        pytest.skip("this scene has a texture or a material that is not regular")
    assert not full.plain_shading()
    img_p, st_p = plain.render(cam)
    img_f, st_f = full.render(cam)
    assert np.array_equal(img_p.view(np.uint32), img_f.view(np.uint32))
    assert (st_p.primary_rays, st_p.secondary_rays, st_p.shadow_rays, st_p.shadow_rays_traced) == (
        st_f.primary_rays, st_f.secondary_rays, st_f.shadow_rays, st_f.shadow_rays_traced)
    plain.set_counters(True)                      # the counted render uses the same kernel: its counts are the oracle's
    _, st_c = plain.render(cam)
    ref, st_o = Oracle(sc).render(cam)
    assert (st_c.primary_rays, st_c.secondary_rays, st_c.shadow_rays, st_c.nodes_visited, st_c.triangle_tests, st_c.shaded_hits) == (
        st_o.primary_rays, st_o.secondary_rays, st_o.shadow_rays, st_o.nodes_visited, st_o.triangle_tests, st_o.shaded_hits)
