"""Largest relative error of a pixel value, GPU against the CPU oracle, per fixture scene and render mode (counted /
timed): the numbers behind the outlier ceilings asserted in tests/test_gpu_parity.py and tests/test_gpu_timed_path.py."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as R
from golden import scenes as GS
from oracle_lib import Oracle
from test_oracle_vs_reference import scene_path

tmp = tempfile.mkdtemp()
for name in ["simple", "mesh70k"] + sorted(GS.SCENES):
    sc = R.Scene(scene_path(name, tmp, ROOT)); cam = sc.camera(0)
    ref, _ = Oracle(sc).render(cam)
    out = [name]
    for mode in ("counted", "timed"):
        if mode == "timed":
            os.environ["RT_WIDE_NODES"] = "1"
            sc = R.Scene(scene_path(name, tmp, ROOT)); cam = sc.camera(0)
        ren = R.Renderer(sc)
        ren.set_counters(mode == "counted")
        img, _ = ren.render(cam)
        err = (np.abs(img - ref) / np.maximum(np.abs(ref), 1 / 255)).max(axis=2)
        out.append(f"{mode}: max {err.max():.3g} over2e-4 {(err > 2e-4).sum()}/{err.size} over1e-5 {(err > 1e-5).sum()}")
    os.environ.pop("RT_WIDE_NODES", None)
    print("  ".join(out), flush=True)
