"""ctypes binding of oracle/liboracle.so — the CPU checker.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

from raytracer_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build()")
        o = C.CDLL(path)
        o.orc_scene_create.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(C.c_void_p)]
        o.orc_scene_destroy.argtypes = [C.c_void_p]
        o.orc_render.argtypes = [C.c_void_p, C.POINTER(A.CameraDesc), C.c_void_p, C.POINTER(A.Stats), C.c_int]
        o.orc_trace_primary.argtypes = [C.c_void_p, C.POINTER(A.CameraDesc), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(A.Stats)]
        o.orc_intersect.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.c_float,
                                    C.c_int, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        o.orc_intersect_wide.argtypes = o.orc_intersect.argtypes
        o.orc_intersect_many.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        o.orc_set_wide.argtypes = [C.c_int]
        o.orc_walk_mismatches.restype = C.c_long
        o.orc_wide_check.restype = C.c_long
        o.orc_wide_check.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        o.orc_math.restype = C.c_float
        o.orc_math.argtypes = [C.c_int, C.c_float, C.c_float]
        o.orc_sample_key.restype = C.c_uint32
        o.orc_sample_key.argtypes = [C.c_uint32] * 3
        o.orc_child_key.restype = C.c_uint32
        o.orc_child_key.argtypes = [C.c_uint32] * 2
        o.orc_rnd.restype = C.c_float
        o.orc_rnd.argtypes = [C.c_uint32] * 2
        _lib = o
    return _lib


class Oracle:
    def __init__(self, scene):
        self._o = lib()
        self.scene = scene
        self._ptr = C.c_void_p()
        if self._o.orc_scene_create(C.byref(scene.desc), C.byref(self._ptr)) != 0:
            raise RuntimeError("orc_scene_create failed")

    def render(self, cam, threads=None):
        out = np.zeros((cam.resy, cam.resx, 3), dtype=np.float32)
        st = A.Stats()
        threads = threads or min(16, os.cpu_count() or 1)
        if self._o.orc_render(self._ptr, C.byref(cam), out.ctypes.data, C.byref(st), threads) != 0:
            raise RuntimeError("orc_render failed")
        return out, st

    def trace_primary(self, cam):
        n = cam.resx * cam.resy
        shape = np.empty(n, dtype=np.int32)
        ids = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        st = A.Stats()
        self._o.orc_trace_primary(self._ptr, C.byref(cam), shape.ctypes.data, ids.ctypes.data, t.ctypes.data, C.byref(st))
        return shape, ids, t, st

    def intersect(self, o, d, time=0.0, culling=False, eps=1e-6, any_hit=False, stop_t=1e30):
        oo = (C.c_float * 3)(*o)
        dd = (C.c_float * 3)(*d)
        s, i, t = C.c_int32(), C.c_int32(), C.c_float()
        hit = self._o.orc_intersect(self._ptr, oo, dd, time, int(culling), eps, int(any_hit), stop_t, C.byref(s), C.byref(i),
                                    C.byref(t))
        return bool(hit), s.value, i.value, t.value

    def intersect_many(self, origins, directions, stop_t=None, any_hit=False, eps=1e-6, wide=False):
        """Rays (n x 3 float32 each) through the two-child walk or, with `wide`, through the product's wide tree walked on
        the host; returns shape (-1: none), instance and distance per ray."""
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        directions = np.ascontiguousarray(directions, dtype=np.float32)
        n = len(origins)
        stop = np.full(n, 1e30, np.float32) if stop_t is None else np.ascontiguousarray(stop_t, dtype=np.float32)
        shape, inst, t = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32)
        rc = self._o.orc_intersect_many(self._ptr, int(wide), n, origins.ctypes.data, directions.ctypes.data, stop.ctypes.data,
                                        int(any_hit), eps, shape.ctypes.data, inst.ctypes.data, t.ctypes.data)
        if rc != 0:
            raise RuntimeError("the scene has no wide tree" if rc < 0 else "orc_intersect_many failed")
        return shape, inst, t

    def wide_check(self):
        """(violations, records) of the wide tree against the two-child tree; violations -1: no wide tree."""
        n = C.c_long(0)
        bad = self._o.orc_wide_check(self._ptr, C.byref(n))
        return bad, n.value

    def render_comparing_walks(self, cam):
        """Renders with every ray sent through both walks; returns the image and the number of rays on which the walks
        disagreed (hit, shape, instance or distance bits)."""
        before = self._o.orc_walk_mismatches()
        if self.wide_check()[0] < 0:
            raise RuntimeError("the scene has no wide tree")
        self._o.orc_set_wide(2)
        try:
            img, st = self.render(cam, threads=1)
        finally:
            self._o.orc_set_wide(0)
        return img, self._o.orc_walk_mismatches() - before

    def close(self):
        if self._ptr:
            self._o.orc_scene_destroy(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
