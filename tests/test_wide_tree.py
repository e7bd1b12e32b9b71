"""The wide tree of the trace kernel (raytracer_amd/csrc/gpu/rt_wide.h) on the host: the records are sound with
respect to the two-child tree, and a walk of them -- the device's step restated in the checker -- finds exactly what the
reference's walk (src/bvh.cpp:383-480) finds: same primitive, same distance bits, closest hit and any hit."""
import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import scenegen
from golden import scenes as GS
from oracle_lib import Oracle

WIDE_SCENES = ["materials", "transforms", "pathtrace", "distributed", "emitters"]


def scene_of(name, scene_dir):
    if name == "mesh70k":
        return R.Scene(scenegen.write_mesh_scene(scene_dir + "/mesh"))
    return R.Scene(GS.SCENES[name][0](scene_dir))


@pytest.mark.parametrize("name", WIDE_SCENES + ["mesh70k"])
def test_records_contain_the_exact_boxes(name, scene_dir):
    bad, records = Oracle(scene_of(name, scene_dir)).wide_check()
    assert bad == 0 and records > 0


def rays_for(sc, orc, n, rng):
    """Camera-like rays, then rays leaving the points they hit (with a small offset, as shading does)."""
    cam = sc.camera(0)
    eye = np.array(cam.position, np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32) * np.float32(0.35) - np.array(cam.w, np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(eye, (n, 1))
    shape, _, t = orc.intersect_many(o, d)
    hit = shape >= 0
    p = (o + d * t[:, None])[hit]
    d2 = rng.normal(size=p.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    return (o, d), ((p + d2 * np.float32(1e-3)).astype(np.float32), d2)


@pytest.mark.parametrize("name", WIDE_SCENES + ["mesh70k"])
def test_both_walks_find_the_same_hits(name, scene_dir):
    sc = scene_of(name, scene_dir)
    orc = Oracle(sc)
    rng = np.random.default_rng(7)
    primary, secondary = rays_for(sc, orc, 20000, rng)
    assert len(secondary[0]) > 2000
    for o, d in (primary, secondary):
        a = orc.intersect_many(o, d)
        b = orc.intersect_many(o, d, wide=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    # any hit with the shadow-ray epsilon, towards a point above the scene: the first entry found ends the walk, so
    # this also pins the ORDER in which leaves are reached (src/ray_trace.cpp:207-241 relies on it)
    o, _ = secondary
    target = np.array([0.3, 3.4, 0.8], np.float32)
    d = target - o
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    d = (d / dist[:, None]).astype(np.float32)
    a = orc.intersect_many(o, d, stop_t=dist, any_hit=True, eps=1e-3)
    b = orc.intersect_many(o, d, stop_t=dist, any_hit=True, eps=1e-3, wide=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert (a[0] >= 0).any()


@pytest.mark.parametrize("name", ["pathtrace", "distributed"])
def test_a_render_never_sees_the_walks_disagree(name, scene_dir):
    """Every ray of a whole render (camera, bounce, refraction and shadow rays) through both walks.  `pathtrace` is the
    scene where quantised boxes alone would not do: a shadow ray grazes a large triangle within the slack the shadow-ray
    epsilon gives the triangle test, outside the triangle's own box -- the walk has to ask the leaf's exact box."""
    sc = scene_of(name, scene_dir)
    cam = sc.camera(0, num_samples=4)
    img, disagreements = Oracle(sc).render_comparing_walks(cam)
    assert disagreements == 0 and np.isfinite(img).all()


def chain_tree(depth, bad_box=False):
    """A two-child tree that is one long chain: every interior node has a leaf on the left and the rest on the right."""
    import ctypes as C
    from raytracer_amd import _abi as A
    n = 2 * depth + 1
    nodes = (A.BvhNode * n)()
    for d in range(depth):
        i = 2 * d
        lo, hi = float(d), float(depth + 1)
        for node, (a, b) in ((nodes[i], (lo, hi)), (nodes[i + 1], (lo, lo + 1.0))):
            for k in range(3):
                node.bmin[k], node.bmax[k] = a, b
        nodes[i].n_prims, nodes[i].offset, nodes[i].axis = 0, i + 2, d % 3
        nodes[i + 1].n_prims, nodes[i + 1].offset = 1, d
    last = nodes[n - 1]
    for k in range(3):
        last.bmin[k], last.bmax[k] = float(depth), float(depth + 1)
    last.n_prims, last.offset = 1, depth
    if bad_box:
        nodes[3].bmax[1] = float("inf")
    return nodes, n


def test_builder_declines_what_the_walk_cannot_hold():
    """The walk's stack has the reference's 64 entries (src/bvh.cpp:423).  A record leaves up to three entries waiting, so a
    tree whose wide walk could need more than 64 is declined (the scene then renders from the two-box records), as is a
    tree with a box that is not finite."""
    from oracle_lib import lib
    import ctypes as C
    L = lib()
    L.orc_wide_build_raw.restype = C.c_long
    L.orc_wide_build_raw.argtypes = [C.c_void_p, C.c_int, C.c_int]
    nodes, n = chain_tree(40)
    assert L.orc_wide_build_raw(nodes, n, 64) == 20          # two chain links per record
    nodes, n = chain_tree(400)
    assert L.orc_wide_build_raw(nodes, n, 64) == -1          # 200 records deep, one entry waiting at each
    assert L.orc_wide_build_raw(nodes, n, 1000) == 200
    nodes, n = chain_tree(40, bad_box=True)
    assert L.orc_wide_build_raw(nodes, n, 64) == -1
