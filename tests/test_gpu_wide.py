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
def test_wide_walk_equals_two_box_walk(name, scene_dir):
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
