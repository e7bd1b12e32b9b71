#!/usr/bin/env python3
"""Regenerates tests/golden/outputs.npz from the compiled reference executable (oracle/_ref/raytracer).

Run in the CPU container where /root/reference is present:  python tests/golden/make_output_golden.py
The reference renders one EXR scene with three tone-map operators and one plain PNG scene; the fixture holds
  hdr        (H, W, 3) float32  the EXR the reference wrote, rows in file order (= framebuffer order, row 0 = bottom)
  tm_<name>  (H, W, 3) uint8    each tone-mapped PNG it wrote, flipped back to framebuffer order
  tm_params  (3, 5)   float32   (operator, option 1, option 2, saturation, gamma) per tone map
  ldr_hdr / ldr_png             the deterministic `simple` scene's framebuffer (from ref_dump) and the PNG of it
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import scenes as S  # noqa: E402
import raytracer_amd as R  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "raytracer")
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
TONEMAPS = [
    {"TMO": "Photographic", "TMOOptions": "0.18 2", "Saturation": "1.0", "Gamma": "2.2", "Extension": "_phot.png"},
    {"TMO": "Filmic", "TMOOptions": "0.3 1", "Saturation": "0.7", "Gamma": "2.2", "Extension": "_filmic.png"},
    {"TMO": "ACES", "TMOOptions": "0.5 0", "Saturation": "1.0", "Gamma": "1.8", "Extension": "_aces.png"},
]
CODES = {"Photographic": 0, "Filmic": 1, "ACES": 2}


def read_exr(path):
    h = R.host_lib()
    w, hh = C.c_int32(), C.c_int32()
    assert h.rth_read_exr(os.fsencode(path), C.byref(w), C.byref(hh), None) == 0, h.rth_last_error()
    rgba = np.empty((hh.value, w.value, 4), dtype=np.float32)
    assert h.rth_read_exr(os.fsencode(path), C.byref(w), C.byref(hh), rgba.ctypes.data) == 0
    return rgba[:, :, :3].copy()


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        path = S.SCENES["envmap"][0](os.path.join(tmp, "scenes"))
        doc = json.load(open(path))
        doc["Scene"]["Cameras"]["Camera"]["Tonemap"] = TONEMAPS
        json.dump(doc, open(path, "w"))
        subprocess.check_call([REF, path, "0", "4"], cwd=tmp, stdout=subprocess.DEVNULL)
        out["hdr"] = read_exr(os.path.join(tmp, "outputs", "envmap.exr"))
        params = []
        for tm in TONEMAPS:
            png = np.asarray(Image.open(os.path.join(tmp, "outputs", "envmap" + tm["Extension"])).convert("RGB"))
            out["tm" + tm["Extension"][:-4]] = png[::-1].copy()
            o1, o2 = [float(x) for x in tm["TMOOptions"].split()]
            params.append([CODES[tm["TMO"]], o1, o2, float(tm["Saturation"]), float(tm["Gamma"])])
        out["tm_params"] = np.array(params, dtype=np.float32)

        simple = os.path.join(ROOT, "scenes", "simple.json")
        subprocess.check_call([REF, simple, "0", "4"], cwd=tmp, stdout=subprocess.DEVNULL)
        name = json.load(open(simple))["Scene"]["Cameras"]["Camera"]["ImageName"]
        out["ldr_png"] = np.asarray(Image.open(os.path.join(tmp, "outputs", name)).convert("RGB"))[::-1].copy()
        prefix = os.path.join(tmp, "simple")
        subprocess.check_call([REF_DUMP, simple, prefix], cwd=tmp, stdout=subprocess.DEVNULL)
        w, h = [int(x) for x in open(prefix + ".meta.txt").read().split()[:2]]
        out["ldr_hdr"] = np.fromfile(prefix + ".hdr.f32", dtype=np.float32).reshape(h, w, 3)
    np.savez_compressed(os.path.join(HERE, "outputs.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
