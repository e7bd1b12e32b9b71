"""Study script (not a test; run by hand): what the two walks of the checker do per ray on the reduced 1M-triangle scene,
and a log of the records and leaves each ray of the wide walk touches for tests/studies/wide_cache_sim.cpp.

  python tests/studies/wide_walk_study.py work          per-ray steps, leaves reached / passed, pushes, deepest stack
  python tests/studies/wide_walk_study.py log OUT_DIR   writes OUT_DIR/widelog.npy and OUT_DIR/widerefs.npy
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import raytracer_amd as R
from raytracer_amd import scenegen
import oracle_lib


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "work"
    L = oracle_lib.lib()
    if what == "work":
        sc = R.Scene(scenegen.write_path_traced_scene("/tmp/wide_study", width=240, height=135, spp=4))
        cam = sc.camera(0)
        o = oracle_lib.Oracle(sc)
        assert o.wide_check()[0] == 0
        w = (C.c_long * 6)()
        L.orc_walk_work(w)
        for mode in (0, 1, 2):   # 2: the wide walk with the slots' exact boxes -- what the 8-bit grid costs in steps
            L.orc_set_exact_slots(1 if mode == 2 else 0)
            L.orc_set_wide(min(mode, 1))
            _, st = o.render(cam, threads=1)
            L.orc_set_wide(0)
            L.orc_walk_work(w)
            rays = st.primary_rays + st.shadow_rays + st.secondary_rays
            L.orc_set_exact_slots(0)
            print(("two-child walk", "wide walk     ", "wide, exact   ")[mode], f"{rays} rays; per ray: interior nodes expanded {w[4] / rays:.2f}, wide steps {w[0] / rays:.2f}, "
                  f"leaves reached {w[1] / rays:.2f}, leaves whose exact box passed {w[5] / rays:.2f}, pushes {w[2] / rays:.2f}, "
                  f"deepest stack {w[3]}, triangle tests {st.triangle_tests / rays:.2f}")
    else:
        out = sys.argv[2]
        os.makedirs(out, exist_ok=True)
        sc = R.Scene(scenegen.write_path_traced_scene("/tmp/wide_study", width=160, height=90, spp=4))
        cam = sc.camera(0)
        o = oracle_lib.Oracle(sc)
        L.orc_wide_records.restype = C.c_long
        L.orc_wide_records.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        n = L.orc_wide_records(o._ptr, None, 0)
        rec = np.empty((n, 16), np.float32)
        L.orc_wide_records(o._ptr, rec.ctypes.data, n)
        np.save(os.path.join(out, "widerefs.npy"), np.ascontiguousarray(rec.view(np.int32)[:, 4:8]))
        room = 200_000_000
        buf = np.empty(room, np.int32)
        L.orc_wide_log.restype = C.c_long
        L.orc_wide_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        m = L.orc_wide_log(o._ptr, C.byref(cam), buf.ctypes.data, room)
        np.save(os.path.join(out, "widelog.npy"), buf[:m].copy())
        print(n, "records,", m, "log entries")


if __name__ == "__main__":
    main()
