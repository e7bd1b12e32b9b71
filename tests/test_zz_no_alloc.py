"""rt_tonemap_noalloc -- the tone mapper for callers that cannot allocate device memory -- against rt_tonemap, the entry
point it is a variant of, through the C ABI.  Both run the same kernels; the variant keeps its partial sums, selection
state and histogram in a fixed device array inside the library, so the bytes must be identical.  Two calls in a row,
the second on a smaller frame: whatever the first call left in the array must not leak into the second."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import _abi as A

pytestmark = pytest.mark.gpu


class Hip:
    """Device buffers through the HIP runtime the rendering core itself is linked against (the copy already mapped into
    this process; a second HIP runtime in the same process -- torch ships its own -- does not find the GPU)."""

    def __init__(self):
        R.gpu_lib()
        path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)
        self.rt = C.CDLL(path)
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.owned = []

    def upload(self, array):
        array = np.ascontiguousarray(array)
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(array.nbytes, 1)) == 0
        self.owned.append(p)
        assert self.rt.hipMemcpy(p, array.ctypes.data, array.nbytes, 1) == 0      # host to device
        return p

    def filled(self, nbytes, byte):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), nbytes) == 0
        self.owned.append(p)
        assert self.rt.hipMemset(p, byte, nbytes) == 0
        return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert self.rt.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0            # device to host
        return out

    def close(self):
        for p in self.owned:
            self.rt.hipFree(p)
        self.owned = []


@pytest.fixture
def hip():
    h = Hip()
    yield h
    h.close()


def frame(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.gamma(0.8, 1.5, size=(h, w, 3)).astype(np.float32)
    img[rng.random((h, w)) < 0.02] *= 40.0          # a few very bright pixels: the burn-percent white point matters
    img[rng.random((h, w)) < 0.01] = 0.0            # and some black ones (luminance below the 1e-6 guard)
    return img


def test_noalloc_tonemap_equals_tonemap(hip):
    g = R.gpu_lib()
    g.rt_tonemap_noalloc.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(A.TonemapDesc), C.c_void_p]
    cases = [((360, 640), A.TonemapDesc(0, 0.18, 2.0, 1.0, 2.2)),      # photographic with a burn percentage
             ((100, 200), A.TonemapDesc(0, 0.25, 5.0, 0.8, 1.8)),      # smaller frame right after, other rank
             ((100, 200), A.TonemapDesc(2, 0.5, 0.0, 1.0, 2.2)),       # ACES, no white-point selection
             ((33, 7), A.TonemapDesc(1, 0.18, 1.0, 1.2, 0.0))]         # filmic, tiny ragged frame, default gamma
    for k, ((h, w), desc) in enumerate(cases):
        hdr = hip.upload(frame(h, w, 100 + k))
        got_d, want_d = hip.filled(h * w * 3, 7), hip.filled(h * w * 3, 9)
        assert g.rt_tonemap_noalloc(hdr, w, h, C.byref(desc), got_d) == 0, g.rt_last_error()
        assert g.rt_tonemap(hdr, w, h, C.byref(desc), want_d) == 0, g.rt_last_error()
        got, want = hip.download(got_d, (h, w, 3), np.uint8), hip.download(want_d, (h, w, 3), np.uint8)
        assert np.array_equal(got, want), f"case {k}: {(got != want).sum()} bytes differ"
        assert want.astype(np.float32).std() > 10      # a real image, not a constant
        host_want = np.empty((h, w, 3), dtype=np.uint8)  # and the existing entry point on host buffers agrees
        assert g.rt_tonemap(frame(h, w, 100 + k).ctypes.data, w, h, C.byref(desc), host_want.ctypes.data) == 0
        assert np.array_equal(got, host_want)


def test_noalloc_tonemap_rejects_what_it_cannot_do(hip):
    g = R.gpu_lib()
    g.rt_tonemap_noalloc.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(A.TonemapDesc), C.c_void_p]
    desc = A.TonemapDesc(0, 0.18, 1.0, 1.0, 2.2)
    hdr = hip.upload(np.ones((4, 4, 3), dtype=np.float32))
    out = hip.filled(4 * 4 * 3, 0)
    assert g.rt_tonemap_noalloc(hdr, 1 << 15, 1 << 14, C.byref(desc), out) == 5     # 2^29 pixels
    assert b"RT_TONEMAP_NOALLOC_MAX_PIXELS" in g.rt_last_error()
    host = np.ones((4, 4, 3), dtype=np.float32)
    assert g.rt_tonemap_noalloc(host.ctypes.data, 4, 4, C.byref(desc), out) == 1    # host memory
    assert b"device memory" in g.rt_last_error()
    assert g.rt_tonemap_noalloc(hdr, 4, 4, C.byref(desc), out) == 0
    # memory of another GPU: the kernels would run on the current device with its own scratch array (a box with one GPU
    # has nothing to try this on)
    count = C.c_int(0)
    assert hip.rt.hipGetDeviceCount(C.byref(count)) == 0
    if count.value > 1:
        assert hip.rt.hipSetDevice(1) == 0
        far = hip.filled(4 * 4 * 3, 0)
        assert hip.rt.hipSetDevice(0) == 0
        assert g.rt_tonemap_noalloc(hdr, 4, 4, C.byref(desc), far) == 1
        assert b"current device" in g.rt_last_error()
