"""Python bindings (ctypes) of the MI355X rendering core and of its host library.

The product is the C-ABI shared library ``librt_gpu.so`` (include/gpu_render.h) plus the C++ host
(``librt_host.so``, the ``raytracer`` CLI).  This module is the thin binding used by the tests,
``bench.py`` and scripting; it mirrors the reference's call sequence in process_file()
(reference src/main.cpp:347-633): load scene -> create tracer -> render each camera -> write images.

There is no CPU fallback: constructing a :class:`Renderer` without the HIP library raises.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_host = None
_gpu = None


class RaytracerError(RuntimeError):
    pass


def host_lib():
    """Loads librt_host.so (scene ingest, BVH build, tone mapping, image files)."""
    global _host
    if _host is None:
        path = os.path.join(_HERE, "librt_host.so")
        if not os.path.exists(path):
            raise RaytracerError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        h = C.CDLL(path)
        h.rth_scene_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        h.rth_scene_desc.restype = C.POINTER(A.SceneDesc)
        h.rth_scene_desc.argtypes = [C.c_void_p]
        h.rth_scene_camera_count.argtypes = [C.c_void_p]
        h.rth_scene_camera.restype = C.POINTER(A.CameraDesc)
        h.rth_scene_camera.argtypes = [C.c_void_p, C.c_int32]
        h.rth_scene_image_name.restype = C.c_char_p
        h.rth_scene_image_name.argtypes = [C.c_void_p, C.c_int32]
        h.rth_scene_tonemap_count.argtypes = [C.c_void_p, C.c_int32]
        h.rth_scene_tonemap_extension.restype = C.c_char_p
        h.rth_scene_tonemap_extension.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        h.rth_scene_build_bvh.argtypes = [C.c_void_p]
        h.rth_scene_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        h.rth_scene_free.argtypes = [C.c_void_p]
        h.rth_tonemap.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p]
        h.rth_scene_tonemap_apply.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        h.rth_quantize_ldr.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
        h.rth_write_png.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        h.rth_write_exr.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]
        h.rth_read_exr.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]
        h.rth_decode_image_rgb8.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]
        h.rth_last_error.restype = C.c_char_p
        _host = h
    return _host


def gpu_lib():
    """Loads librt_gpu.so (the HIP rendering core).  Fails loudly when it is missing."""
    global _gpu
    if _gpu is None:
        path = os.path.join(_HERE, "librt_gpu.so")
        if not os.path.exists(path):
            raise RaytracerError(f"{path} is missing: the HIP rendering core has not been built "
                                 "(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        g = C.CDLL(path)
        g.rt_scene_create.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(C.c_void_p)]
        g.rt_render.argtypes = [C.c_void_p, C.POINTER(A.CameraDesc), C.c_void_p, C.POINTER(A.Stats)]
        g.rt_render_shard.argtypes = [C.c_void_p, C.POINTER(A.CameraDesc), C.c_int32, C.c_int32, C.c_void_p,
                                      C.POINTER(A.Stats)]
        g.rt_trace_primary.argtypes = [C.c_void_p, C.POINTER(A.CameraDesc), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(A.Stats)]
        g.rt_shard_owner_map.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        g.rt_shard_layout.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        g.rt_comm_unique_id.argtypes = [C.c_void_p, C.c_int32]
        g.rt_comm_init_rank.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        g.rt_comm_init_all.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
        g.rt_render_gather.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(A.CameraDesc), C.c_void_p, C.POINTER(A.Stats)]
        g.rt_comm_destroy.argtypes = [C.c_void_p]
        g.rt_set_counters.argtypes = [C.c_void_p, C.c_int32]
        g.rt_tonemap.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(A.TonemapDesc), C.c_void_p]
        g.rt_tonemap_noalloc.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(A.TonemapDesc), C.c_void_p]
        g.rt_quantize_ldr.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        g.rt_set_profiling.argtypes = [C.c_void_p, C.c_int32]
        g.rt_kernel_times.argtypes = [C.c_void_p, C.POINTER(A.KernelTime)]
        g.rt_scene_destroy.argtypes = [C.c_void_p]
        g.rt_last_error.restype = C.c_char_p
        _gpu = g
    return _gpu


class Scene:
    """A parsed scene file: flat arrays + cameras (host memory)."""

    def __init__(self, json_path):
        self._h = host_lib()
        self._ptr = C.c_void_p()
        rc = self._h.rth_scene_load(os.fsencode(json_path), C.byref(self._ptr))
        if rc != 0:
            raise RaytracerError(f"cannot load {json_path}: {self._h.rth_last_error().decode()}")
        self.path = json_path
        self.desc = self._h.rth_scene_desc(self._ptr).contents

    @property
    def n_cameras(self):
        return self._h.rth_scene_camera_count(self._ptr)

    def camera(self, index=0, **overrides):
        """A copy of camera `index`'s description; keyword arguments override fields (e.g. seed=1)."""
        src = self._h.rth_scene_camera(self._ptr, index)
        if not src:
            raise IndexError(index)
        cam = A.CameraDesc()
        C.memmove(C.byref(cam), src, C.sizeof(A.CameraDesc))
        for k, v in overrides.items():
            setattr(cam, k, v)
        return cam

    def image_name(self, index=0):
        return self._h.rth_scene_image_name(self._ptr, index).decode()

    def tonemap_count(self, camera=0):
        return self._h.rth_scene_tonemap_count(self._ptr, camera)

    def tonemap_extension(self, camera, index):
        return self._h.rth_scene_tonemap_extension(self._ptr, camera, index).decode()

    def tonemap(self, camera, index, hdr):
        hdr = np.ascontiguousarray(hdr, dtype=np.float32)
        out = np.empty(hdr.shape, dtype=np.uint8)
        if self._h.rth_scene_tonemap_apply(self._ptr, camera, index, hdr.ctypes.data, out.ctypes.data) != 0:
            raise RaytracerError("bad tonemap index")
        return out

    def build_bvh(self):
        """Builds the host's SAH tree of a scene that was loaded without one (RT_GPU_BVH=1)."""
        self._h.rth_scene_build_bvh(self._ptr)
        self.desc = self._h.rth_scene_desc(self._ptr).contents

    def timings(self):
        a, b = C.c_double(), C.c_double()
        self._h.rth_scene_timings(self._ptr, C.byref(a), C.byref(b))
        return {"ingest_ms": a.value, "bvh_ms": b.value}

    def bvh_arrays(self):
        d = self.desc
        nodes = np.ctypeslib.as_array(C.cast(d.bvh_nodes, C.POINTER(C.c_int32)), shape=(d.n_bvh_nodes, 9)).copy()
        bounds = nodes[:, :6].copy().view(np.float32)
        prims = np.ctypeslib.as_array(C.cast(d.bvh_prims, C.POINTER(C.c_int32)), shape=(d.n_bvh_prims, 2)).copy()
        return bounds, nodes[:, 6:9].copy(), prims

    def close(self):
        if self._ptr:
            self._h.rth_scene_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """Scene resident on the current HIP device (rt_scene_create ... rt_scene_destroy)."""

    def __init__(self, scene):
        self._g = gpu_lib()
        self.scene = scene
        self._ptr = C.c_void_p()
        if self._g.rt_scene_create(C.byref(scene.desc), C.byref(self._ptr)) != 0:
            raise RaytracerError("rt_scene_create: " + self._g.rt_last_error().decode())

    def _check(self, rc, what):
        if rc != 0:
            raise RaytracerError(f"{what}: {self._g.rt_last_error().decode()}")

    def shadow_order_free(self):
        """True when the scene's shadow rays are walked in any order (rt_scene_shadow_order_free)."""
        self._g.rt_scene_shadow_order_free.restype = C.c_int32
        self._g.rt_scene_shadow_order_free.argtypes = [C.c_void_p]
        return bool(self._g.rt_scene_shadow_order_free(self._ptr))

    def plain_shading(self):
        """True when the scene is shaded by the shading kernel built without textures and mirrors (rt_scene_plain_shading)."""
        self._g.rt_scene_plain_shading.restype = C.c_int32
        self._g.rt_scene_plain_shading.argtypes = [C.c_void_p]
        return bool(self._g.rt_scene_plain_shading(self._ptr))

    def wide_records(self):
        """True when the scene holds the four-box records that uncounted frames walk (rt_scene_wide_records)."""
        self._g.rt_scene_wide_records.restype = C.c_int32
        self._g.rt_scene_wide_records.argtypes = [C.c_void_p]
        return bool(self._g.rt_scene_wide_records(self._ptr))

    def set_counters(self, enabled):
        self._check(self._g.rt_set_counters(self._ptr, 1 if enabled else 0), "rt_set_counters")

    def set_profiling(self, enabled):
        """Brackets every kernel launch of later renders with HIP events; clears the totals."""
        self._check(self._g.rt_set_profiling(self._ptr, 1 if enabled else 0), "rt_set_profiling")

    def kernel_times(self):
        """{kernel name: (launches, total ms)} since set_profiling."""
        arr = (A.KernelTime * len(A.KERNEL_NAMES))()
        self._check(self._g.rt_kernel_times(self._ptr, arr), "rt_kernel_times")
        return {n: (int(arr[i].launches), float(arr[i].total_ms)) for i, n in enumerate(A.KERNEL_NAMES)}

    def render(self, cam, out=None, rank=0, world=1):
        """Renders `cam`; returns (hdr[resy, resx, 3] float32 with row 0 = bottom row, Stats).
        `out` may be a numpy array or an integer device/host pointer."""
        stats = A.Stats()
        if out is None:
            out = np.zeros((cam.resy, cam.resx, 3), dtype=np.float32)
        ptr = out if isinstance(out, int) else out.ctypes.data
        if world == 1:
            self._check(self._g.rt_render(self._ptr, C.byref(cam), ptr, C.byref(stats)), "rt_render")
        else:
            self._check(self._g.rt_render_shard(self._ptr, C.byref(cam), rank, world, ptr, C.byref(stats)), "rt_render_shard")
        return out, stats

    def render_gather(self, comm, cam, out=None):
        """This rank's part of a frame that is split over the ranks of `comm` (rt_render_gather).  On rank 0 returns
        (frame, Stats) like render(); on the other ranks (None, Stats).  Every rank of the communicator must call it."""
        stats = A.Stats()
        ptr = None
        if comm.rank == 0:
            if out is None:
                out = np.zeros((cam.resy, cam.resx, 3), dtype=np.float32)
            ptr = out if isinstance(out, int) else out.ctypes.data
        self._check(self._g.rt_render_gather(self._ptr, comm._ptr, C.byref(cam), ptr, C.byref(stats)), "rt_render_gather")
        return (out if comm.rank == 0 else None), stats

    def trace_primary(self, cam):
        n = cam.resx * cam.resy
        shape = np.empty(n, dtype=np.int32)
        ids = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        stats = A.Stats()
        self._check(self._g.rt_trace_primary(self._ptr, C.byref(cam), shape.ctypes.data, ids.ctypes.data, t.ctypes.data,
                                             C.byref(stats)), "rt_trace_primary")
        return shape, ids, t, stats

    def close(self):
        if self._ptr:
            self._g.rt_scene_destroy(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """One rank's end of a communicator for tile-sharded frames (rt_comm_*)."""

    def __init__(self, ptr, rank, world):
        self._g = gpu_lib()
        self._ptr = C.c_void_p(ptr)
        self.rank, self.world = rank, world

    @classmethod
    def init_all(cls, devices):
        """Ranks as threads of this process, rank r on devices[r]; ranks that share a device (a rehearsal on one GPU)
        exchange by device-to-device copies.  Returns one Comm per rank."""
        g = gpu_lib()
        n = len(devices)
        devs = (C.c_int32 * n)(*devices)
        ptrs = (C.c_void_p * n)()
        if g.rt_comm_init_all(n, devs, ptrs) != 0:
            raise RaytracerError("rt_comm_init_all: " + g.rt_last_error().decode())
        return [cls(ptrs[r], r, n) for r in range(n)]

    @classmethod
    def init_rank(cls, unique_id, rank, world):
        """This process as rank `rank` of `world` (one process per GPU, the current device); `unique_id` from
        sharding.exchange_unique_id."""
        g = gpu_lib()
        ptr = C.c_void_p()
        buf = (C.c_ubyte * len(unique_id)).from_buffer_copy(unique_id) if unique_id else None
        if g.rt_comm_init_rank(buf, len(unique_id) if unique_id else 0, rank, world, C.byref(ptr)) != 0:
            raise RaytracerError("rt_comm_init_rank: " + g.rt_last_error().decode())
        return cls(ptr.value, rank, world)

    def close(self):
        if self._ptr:
            self._g.rt_comm_destroy(self._ptr)
            self._ptr = C.c_void_p()


def gpu_tonemap(hdr, tmo, opt1, opt2, saturation, gamma):
    """rt_tonemap on a host array (H, W, 3) float32 -> uint8 array of the same shape."""
    hdr = np.ascontiguousarray(hdr, dtype=np.float32)
    out = np.empty(hdr.shape, dtype=np.uint8)
    desc = A.TonemapDesc(int(tmo), float(opt1), float(opt2), float(saturation), float(gamma))
    g = gpu_lib()
    if g.rt_tonemap(hdr.ctypes.data, hdr.shape[1], hdr.shape[0], C.byref(desc), out.ctypes.data) != 0:
        raise RaytracerError("rt_tonemap: " + g.rt_last_error().decode())
    return out


def gpu_quantize_ldr(hdr):
    hdr = np.ascontiguousarray(hdr, dtype=np.float32)
    out = np.empty(hdr.shape, dtype=np.uint8)
    g = gpu_lib()
    if g.rt_quantize_ldr(hdr.ctypes.data, hdr.size, out.ctypes.data) != 0:
        raise RaytracerError("rt_quantize_ldr: " + g.rt_last_error().decode())
    return out


def write_outputs(scene, camera_index, hdr, out_dir="outputs"):
    """Writes the images the reference writes for one camera (src/main.cpp:561-607): an EXR when the camera is HDR or the
    image name ends in .exr, else a plain PNG when the camera has no tone maps; and one PNG per tone map
    (`<base><Extension>`).  Returns the paths written."""
    h = host_lib()
    os.makedirs(out_dir, exist_ok=True)
    name = scene.image_name(camera_index)
    hdr = np.ascontiguousarray(hdr, dtype=np.float32)
    H, W = hdr.shape[:2]
    written = []
    ntm = scene.tonemap_count(camera_index)
    if name.endswith(".exr") or scene.camera(camera_index).is_hdr:
        path = os.path.join(out_dir, name)
        h.rth_write_exr(os.fsencode(path), W, H, hdr.ctypes.data)
        written.append(path)
    elif ntm == 0:
        ldr = np.empty(hdr.shape, dtype=np.uint8)
        h.rth_quantize_ldr(hdr.size, hdr.ctypes.data, ldr.ctypes.data)
        path = os.path.join(out_dir, name)
        h.rth_write_png(os.fsencode(path), W, H, ldr.ctypes.data, 1)
        written.append(path)
    base = name[:name.rfind(".")] if "." in name else name
    for k in range(ntm):
        ldr = scene.tonemap(camera_index, k, hdr)
        path = os.path.join(out_dir, base + scene.tonemap_extension(camera_index, k))
        h.rth_write_png(os.fsencode(path), W, H, ldr.ctypes.data, 1)
        written.append(path)
    return written
