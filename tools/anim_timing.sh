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
for k in range(40):
    cam = doc["Scene"]["Cameras"]["Camera"]
    a = 2 * math.pi * k / 40
    cam["Position"] = f"{3.2 * math.sin(a):.5f} 0.5 {3.2 * math.cos(a):.5f}"
    cam["ImageName"] = f"anim_{k:04d}.png"
    json.dump(doc, open(os.path.join(frames, f"anim_{k:04d}.json"), "w"))
P
cd "$W"
RT_GPU_BVH=${RT_GPU_BVH:-1} RT_CLI_TIMING=1 "$ROOT/raytracer" "$W/frames" 2>&1 | grep -i -E "timing|ms" | tail -${LINES_SHOWN:-60}
rm -rf "$W"
