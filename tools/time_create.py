"""rt_scene_create with the host's tree against a tree built on the device (RT_GPU_BVH=1): load and create times of the
70 K-triangle and the 1 M-triangle scene."""
import os, sys, time
sys.path.insert(0, '.')
import raytracer_amd as R
from raytracer_amd import scenegen
for name, path in (("70K", scenegen.write_mesh_scene('/tmp/scn_tc')), ("1M", scenegen.write_path_traced_scene('/tmp/scn_tc', width=960, height=540, spp=4))):
    for mode in ("host tree", "device tree"):
        if mode == "device tree":
            os.environ["RT_GPU_BVH"] = "1"
        else:
            os.environ.pop("RT_GPU_BVH", None)
        best = None
        for rep in range(3):
            t0 = time.perf_counter(); sc = R.Scene(path); t1 = time.perf_counter(); ren = R.Renderer(sc); t2 = time.perf_counter()
            cam = sc.camera(0)
            img, st = ren.render(cam); ren.render(cam)
            img, st = ren.render(cam)
            cur = (t1 - t0, t2 - t1, st.kernel_ms)
            best = cur if best is None or cur[0] + cur[1] < best[0] + best[1] else best
            ren.close()
        print(f"{name:4s} {mode:12s} load {best[0]*1e3:7.1f} ms (ingest {sc.timings()['ingest_ms']:.1f}, bvh {sc.timings()['bvh_ms']:.1f})  create {best[1]*1e3:7.1f} ms  frame {best[2]:.2f} ms")
