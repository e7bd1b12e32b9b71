"""GPU tone mapping and LDR quantisation (rt_tonemap, rt_quantize_ldr) against the host implementation of the same
reference routine (apply_tonemap, src/main.cpp:267-345) and against PNG files the reference executable wrote.

Tolerance: 8-bit outputs may differ by one level where powf/logf of the two maths libraries round a value to the other
side of x.5 -- at most 0.1 % of the values against the host code (0.5 % against the -ffast-math reference binary),
never by more than one level.  The white point is an exact order statistic on both sides."""
import os

import numpy as np
import pytest

import raytracer_amd as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "outputs.npz"))


def host_tonemap(hdr, tmo, o1, o2, sat, gamma):
    hdr = np.ascontiguousarray(hdr, dtype=np.float32)
    out = np.empty(hdr.shape, dtype=np.uint8)
    h, w = hdr.shape[:2]
    assert R.host_lib().rth_tonemap(int(tmo), float(o1), float(o2), float(sat), float(gamma), w, h, hdr.ctypes.data, out.ctypes.data) == 0
    return out


def close(a, b, frac):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1, d.max()
    assert (d != 0).mean() <= frac, (d != 0).mean()


@pytest.mark.parametrize("index,name", [(0, "tm_phot"), (1, "tm_filmic"), (2, "tm_aces")])
def test_reference_frame(index, name):
    tmo, o1, o2, sat, gamma = GOLD["tm_params"][index]
    out = R.gpu_tonemap(GOLD["hdr"], tmo, o1, o2, sat, gamma)
    close(out, host_tonemap(GOLD["hdr"], tmo, o1, o2, sat, gamma), 1e-3)
    close(out, GOLD[name], 5e-3)


@pytest.mark.parametrize("tmo", [0, 1, 2])
@pytest.mark.parametrize("burn,sat,gamma", [(0.0, 1.0, 2.2), (1.0, 0.8, 2.2), (5.0, 1.0, 0.0), (50.0, 1.3, 1.0)])
def test_random_frames(tmo, burn, sat, gamma):
    rng = np.random.default_rng(tmo * 10 + int(burn))
    hdr = np.exp(rng.normal(-1.0, 2.5, size=(237, 301, 3))).astype(np.float32)   # five decades of dynamic range
    hdr[5:9, 7:30] = 0.0                                                           # black pixels: the colour ratio is skipped
    hdr[40, 40] = [-0.5, 0.2, 0.1]                                                 # a negative channel
    out = R.gpu_tonemap(hdr, tmo, 0.18, burn, sat, gamma)
    close(out, host_tonemap(hdr, tmo, 0.18, burn, sat, gamma), 1e-3)
    assert out.std() > 20


def test_single_pixel_and_constant_frames():
    one = np.array([[[0.3, 0.6, 0.1]]], dtype=np.float32)
    close(R.gpu_tonemap(one, 0, 0.18, 1.0, 1.0, 2.2), host_tonemap(one, 0, 0.18, 1.0, 1.0, 2.2), 0.0)
    flat = np.full((64, 48, 3), 2.5, dtype=np.float32)
    close(R.gpu_tonemap(flat, 2, 0.5, 3.0, 1.0, 2.2), host_tonemap(flat, 2, 0.5, 3.0, 1.0, 2.2), 0.0)


def test_quantisation_is_exact():
    rng = np.random.default_rng(5)
    hdr = rng.uniform(-0.2, 1.3, size=(97, 131, 3)).astype(np.float32)
    hdr[0, 0] = [0.0, 1.0, 255.9999 / 255.0]
    host = np.empty(hdr.shape, dtype=np.uint8)
    R.host_lib().rth_quantize_ldr(hdr.size, hdr.ctypes.data, host.ctypes.data)
    assert np.array_equal(R.gpu_quantize_ldr(hdr), host)
    assert np.array_equal(R.gpu_quantize_ldr(GOLD["ldr_hdr"]), GOLD["ldr_png"])
