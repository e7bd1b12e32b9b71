"""Device time of every kernel class per path generation (4K / 64 spp frame of the 1M-triangle scene): runs a child with
RT_PROFILE_LAUNCHES=1 and sorts its per-launch lines by their order inside a batch.  usage: python tools/per_generation.py [spp]"""
import os
import subprocess
import sys

CHILD = r'''
import sys
sys.path.insert(0, ".")
import raytracer_amd as R
from raytracer_amd import scenegen
p = scenegen.write_path_traced_scene("/tmp/scn_kt", width=3840, height=2160, spp=int(sys.argv[1]))
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
sys.stderr.write("== timed frame\n")
ren.set_profiling(True)
img, st = ren.render(cam)
sys.stderr.write("== end %d %d %d\n" % (st.primary_rays, st.secondary_rays, st.shadow_rays_traced))
'''
NAMES = {0: "generate", 1: "trace_closest", 2: "shade", 3: "light_setup", 4: "trace_any", 5: "light_gather", 6: "resolve"}


def main():
    spp = sys.argv[1] if len(sys.argv) > 1 else "64"
    env = dict(os.environ, RT_PROFILE_LAUNCHES="1")
    out = subprocess.run([sys.executable, "-c", CHILD, spp], env=env, stderr=subprocess.PIPE, text=True, check=True).stderr
    lines = out.split("== timed frame\n")[1].split("\n")
    per = {}   # (kernel, generation) -> ms
    gen = {}
    for ln in lines:
        if ln.startswith("== end"):
            print("rays: primary %s secondary %s shadow traversals %s" % tuple(ln.split()[2:5]))
        if not ln.startswith("launch kernel"):
            continue
        k = int(ln.split()[2].rstrip(":"))
        ms = float(ln.split()[3])
        if k == 0:   # a batch begins
            gen = {}
        g = gen.get(k, 0)
        gen[k] = g + 1
        per[(k, g)] = per.get((k, g), 0.0) + ms
    for k in sorted(NAMES):
        row = [per.get((k, g), 0.0) for g in range(8)]
        if any(row):
            print("%-14s" % NAMES[k], " ".join("gen%d %7.1f" % (g, v) for g, v in enumerate(row) if v), " total %.1f ms" % sum(row))


main()
