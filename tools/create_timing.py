"""Stage times of rt_scene_create / rt_scene_destroy (RT_CREATE_TIMING) for the 70 K-triangle scene with its tree built on the
device, third creation in the process (code objects loaded, allocator warm)."""
import os, sys, time
sys.path.insert(0, '.')
os.environ["RT_GPU_BVH"] = os.environ.get("RT_GPU_BVH", "1")
import raytracer_amd as R
from raytracer_amd import scenegen
path = scenegen.write_mesh_scene('/tmp/scn_ct', n=188, width=1920, height=1080)
sc = R.Scene(path)
cam = sc.camera(0)
for k in range(4):
    if k == 3:
        os.environ["RT_CREATE_TIMING"] = "1"
    t0 = time.perf_counter()
    ren = R.Renderer(sc)
    t1 = time.perf_counter()
    ren.render(cam)
    t2 = time.perf_counter()
    ren.close()
    t3 = time.perf_counter()
    print(f"creation {k}: create {1e3 * (t1 - t0):.2f} ms, first frame {1e3 * (t2 - t1):.2f} ms, destroy {1e3 * (t3 - t2):.2f} ms", flush=True)
