"""Per-launch kernel times of one frame of the textured 1080p / 64 spp workload (RT_PROFILE_LAUNCHES), batch by batch."""
import os, subprocess, sys
CHILD = r'''
import sys
sys.path.insert(0, ".")
import bench
import raytracer_amd as R
from raytracer_amd import scenegen
w = bench.WORKLOADS["textured_1080p_64spp"]
p = bench.make_scene(scenegen, "/tmp/c3_scene", w, "textured_1080p_64spp")
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
sys.stderr.write("== timed frame\n")
ren.set_profiling(True)
img, st = ren.render(cam)
sys.stderr.write("== end\n")
'''
env = dict(os.environ, RT_PROFILE_LAUNCHES="1", RT_PRINT_BATCHES="1")
out = subprocess.run([sys.executable, "-c", CHILD], env=env, stderr=subprocess.PIPE, text=True, check=True).stderr
lines = out.split("== timed frame\n")[1].split("\n")
batch = -1
per = {}
sizes = []
for ln in lines:
    if ln.startswith("batch:"):
        sizes.append(ln)
    if not ln.startswith("launch kernel"):
        continue
    k = int(ln.split()[2].rstrip(":")); ms = float(ln.split()[3])
    if k == 0:
        batch += 1
    per.setdefault(batch, {}).setdefault(k, []).append(ms)
print("\n".join(sizes))
for b in sorted(per):
    print("batch", b, {k: [round(x, 2) for x in v] for k, v in per[b].items() if k in (3, 4, 5)})
