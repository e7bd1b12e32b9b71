"""Config 5 rehearsal: 120 JSON frames (70K-triangle mesh, moving camera, 1920x1080) through ./raytracer <dir>,
timed end to end, next to the reference executable on the same frames when it is present."""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import scenegen
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 120
work = tempfile.mkdtemp(prefix="rt_anim_")
frames = os.path.join(work, "frames")
base = scenegen.write_mesh_scene(frames, name="anim_0000", n=188, width=1920, height=1080)
doc = json.load(open(base))
import math
for k in range(n):
    cam = doc["Scene"]["Cameras"]["Camera"]
    a = 2 * math.pi * k / n
    cam["Position"] = f"{3.2 * math.sin(a):.5f} 0.5 {3.2 * math.cos(a):.5f}"
    cam["ImageName"] = f"anim_{k:04d}.png"
    json.dump(doc, open(os.path.join(frames, f"anim_{k:04d}.json"), "w"))
EXE = os.environ.get("RT_BENCH_EXE", os.path.join(ROOT, "raytracer"))   # (another build of the command line, for A/B runs)
runs = [("this repository", EXE, {}),
        ("this repository, trees built on the device (RT_GPU_BVH=1)", EXE, {"RT_GPU_BVH": "1"}),
        ("this repository, trees built on the device, two-box records only (RT_GPU_BVH=1 RT_WIDE_NODES=0)", EXE,
         {"RT_GPU_BVH": "1", "RT_WIDE_NODES": "0"}),
        ("reference executable", os.path.join(ROOT, "oracle", "_ref", "raytracer"), {})]
if "--no-reference" in sys.argv:
    runs.pop()
if "--against-old" in sys.argv:   # A/B on one box: a build of an earlier commit (its raytracer and raytracer_amd/*.so copied to ab_old/), three runs each in turn
    old = os.path.join(ROOT, "ab_old", "raytracer")
    runs = [(f"{label} [{which}, run {r}]", exe if which == "new" else old, env) for r in range(3) for label, exe, env in runs[:2] for which in ("old", "new")]
if "--render-threads" in sys.argv:   # several render threads on the one GPU (RT_GPUS_DEVICES names it more than once)
    runs = [(f"{label}, {k} render threads on the GPU", exe, dict(env, RT_GPUS_DEVICES=",".join(["0"] * k))) for label, exe, env in runs[:2] for k in (1, 2, 3, 4)]
for k, (label, exe, env) in enumerate(runs):
    if not os.path.exists(exe):
        continue
    cwd = os.path.join(work, f"run{k}")
    os.makedirs(cwd)
    t0 = time.perf_counter()
    r = subprocess.run([exe, frames], cwd=cwd, capture_output=True, text=True, env=dict(os.environ, **env))
    dt = time.perf_counter() - t0
    pngs = [os.path.join(cwd, "outputs", f) for f in os.listdir(os.path.join(cwd, "outputs")) if f.endswith(".png")]
    done = len(pngs)
    size = sum(os.path.getsize(f) for f in pngs) / max(1, done) / 1e6
    print(f"{label}: {n} frames in {dt:.2f} s = {n / dt:.1f} frames/s ({done} PNGs written, {size:.2f} MB each, exit {r.returncode})")
shutil.rmtree(work, ignore_errors=True)
