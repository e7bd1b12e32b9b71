#!/bin/bash
# Stage times of the command line on a directory of frames (RT_CLI_TIMING): where a frame's time goes.
# Trees built on the device unless RT_GPU_BVH=0 is set.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/rt_anim_t.XXXX)
python3 - "$W" <<'P'
import json, math, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from raytracer_amd import scenegen
frames = os.path.join(sys.argv[1], "frames")
base = scenegen.write_mesh_scene(frames, name="anim_0000", n=188, width=1920, height=1080)
doc = json.load(open(base))
N = int(os.environ.get('FRAMES', '40'))
for k in range(N):
    cam = doc["Scene"]["Cameras"]["Camera"]
    a = 2 * math.pi * k / N
    cam["Position"] = f"{3.2 * math.sin(a):.5f} 0.5 {3.2 * math.cos(a):.5f}"
    cam["ImageName"] = f"anim_{k:04d}.png"
    json.dump(doc, open(os.path.join(frames, f"anim_{k:04d}.json"), "w"))
P
cd "$W"
RT_GPU_BVH=${RT_GPU_BVH:-1} RT_CLI_TIMING=1 "$ROOT/raytracer" "$W/frames" > "$W/log.txt" 2>&1
grep -i -E "timing|ms" "$W/log.txt" | tail -${LINES_SHOWN:-60}
python3 - "$W/log.txt" <<'P'
import re, sys
text = open(sys.argv[1]).read()
for what, pat in (("load", r"\[timing\] load ([0-9.]+) ms"), ("render thread waited", r"render thread waited ([0-9.]+) ms"), ("scene upload", r"scene upload ([0-9.]+) ms"),
                  ("camera total", r"camera total ([0-9.]+) ms"), ("scene total on the render thread", r"scene total on the render thread ([0-9.]+) ms"),
                  ("PNG written in", r"wrote \S+ in ([0-9.]+) ms")):
    v = [float(x) for x in re.findall(pat, text)]
    if v:
        v.sort()
        print(f"{what}: n {len(v)}, mean {sum(v) / len(v):.2f} ms, median {v[len(v) // 2]:.2f}, max {v[-1]:.2f}")
P
rm -rf "$W"
