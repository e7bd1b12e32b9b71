"""Per-kernel device times of one frame of a bench workload (HIP events around every launch), once on one stream
(the default) and once with the direct lighting on a second stream (RT_TWO_STREAMS=1).
    python tools/kernel_times_full.py [width height spp]"""
import os
import subprocess
import sys

sys.path.insert(0, ".")

CHILD = r'''
import sys
sys.path.insert(0, ".")
import raytracer_amd as R
from raytracer_amd import scenegen
w, h, spp = (int(v) for v in sys.argv[1:4])
p = scenegen.write_path_traced_scene("/tmp/scn_kt", width=w, height=h, spp=spp)
sc = R.Scene(p); ren = R.Renderer(sc); cam = sc.camera(0)
ren.render(cam)
ren.set_profiling(True)
img, st = ren.render(cam)
kt = ren.kernel_times()
print("frame kernel_ms %.1f" % st.kernel_ms, {k: (v[0], round(v[1], 1)) for k, v in kt.items()}, "sum %.1f" % sum(v[1] for v in kt.values()))
'''

if __name__ == "__main__":
    args = sys.argv[1:4] if len(sys.argv) >= 4 else ["3840", "2160", "64"]
    for one in (False, True):
        env = dict(os.environ)
        if one:
            env["RT_TWO_STREAMS"] = "1"
        print(("two streams" if one else "one stream") + "".join(f" {k}={v}" for k, v in env.items() if k.startswith("RT_")), flush=True)
        subprocess.run([sys.executable, "-c", CHILD] + args, env=env, check=True)
