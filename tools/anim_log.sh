cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json, os, sys, math
sys.path.insert(0, '.')
from raytracer_amd import scenegen
frames = '/tmp/anim_frames'
base = scenegen.write_mesh_scene(frames, name="anim_0000", n=188, width=1920, height=1080)
doc = json.load(open(base))
n = 120
for k in range(n):
    cam = doc["Scene"]["Cameras"]["Camera"]
    a = 2 * math.pi * k / n
    cam["Position"] = f"{3.2 * math.sin(a):.5f} 0.5 {3.2 * math.cos(a):.5f}"
    cam["ImageName"] = f"anim_{k:04d}.png"
    json.dump(doc, open(os.path.join(frames, f"anim_{k:04d}.json"), "w"))
PY
mkdir -p /tmp/anim_run && cd /tmp/anim_run && time RT_CLI_TIMING=1 $GRAFT_REPO_ROOT/raytracer /tmp/anim_frames > log.txt 2> err.txt; grep "load " log.txt | sort -k3 -n | tail -3; grep "render thread waited" log.txt | sort -k5 -n | tail -4; grep "scene total" log.txt | sort -k12 -n | tail -3; grep "camera total" log.txt | sort -k4 -n | tail -3; grep "file(s) in" log.txt; grep -c timing log.txt; grep -c "rendering done" log.txt; grep "preparation is done" log.txt | awk '{s+=$8} END {print "avg prep ms", s/NR}'; ls -la outputs | head -5; nproc
