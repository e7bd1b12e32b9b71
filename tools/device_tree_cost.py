"""Frame time with the tree built on the device against the host's SAH tree, each through the two-box walk (RT_WIDE_NODES=0)
and through the four-box walk every uncounted frame runs: 1M-triangle scene at 3840x2160 x 16 spp and the 70 K-triangle scene
at 1080p.  Prints kernel ms per frame (second frame) and the create time."""
import os, sys, time
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen

def frame(path, env):
    for k in ("RT_GPU_BVH", "RT_WIDE_NODES", "RT_GPU_BVH_RADIUS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    sc = R.Scene(path)
    ren = R.Renderer(sc)
    t_create = (time.perf_counter() - t0) * 1e3
    cam = sc.camera(0)
    ren.render(cam)
    ren.set_profiling(True)
    _, st = ren.render(cam)
    kt = ren.kernel_times()
    ren.set_profiling(False)
    ren.set_counters(True)        # the walk's work per ray on this tree (a quarter of the pixels)
    _, sc_ = ren.render(sc.camera(0, resx=cam.resx // 2, resy=cam.resy // 2))
    rays = max(1, sc_.total_rays)
    work = f"per ray: {sc_.nodes_visited / rays:.2f} node visits, {sc_.triangle_tests / rays:.2f} triangle tests"
    return st.kernel_ms, t_create, {k: round(v[1], 1) for k, v in kt.items() if k.startswith("trace")}, work

scenes = [("pt1m 4K x 16 spp", scenegen.write_path_traced_scene('/tmp/scn_dt', width=3840, height=2160, spp=16)),
          ("mesh70k 1080p", scenegen.write_mesh_scene('/tmp/scn_dt2'))]
for name, path in scenes:
    base = None
    for label, env in (("host SAH tree, two-box walk", {"RT_WIDE_NODES": "0"}), ("host SAH tree, wide walk", {"RT_WIDE_NODES": "1"}),
                       ("device tree, two-box walk", {"RT_GPU_BVH": "1", "RT_WIDE_NODES": "0"}),
                       ("device tree, wide walk", {"RT_GPU_BVH": "1"}),
                       ) + tuple((f"device tree, wide walk, search radius {r}", {"RT_GPU_BVH": "1", "RT_GPU_BVH_RADIUS": str(r)})
                                 for r in (int(x) for x in os.environ.get("RT_TOOL_RADII", "4,16").split(","))):
        ms, create, kt, work = frame(path, env)
        base = base or ms
        print(f"{name}: {label}: frame {ms:.2f} ms ({ms / base:.3f} of the first), scene load + create {create:.0f} ms, {kt}, {work}", flush=True)
