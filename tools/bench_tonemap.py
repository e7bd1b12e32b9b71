"""Times rt_tonemap on a 3840x2160 frame resident in HBM against the host implementation (apply_tonemap)."""
import ctypes as C
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch

import raytracer_amd as R
from raytracer_amd import _abi as A

W, H = 3840, 2160
rng = np.random.default_rng(0)
hdr = np.exp(rng.normal(-1.0, 2.0, size=(H, W, 3))).astype(np.float32)
d_hdr = torch.from_numpy(hdr).cuda()
d_out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
g = R.gpu_lib()
desc = A.TonemapDesc(0, 0.18, 1.0, 1.0, 2.2)
for label, burn in (("photographic, burn 1 %", 1.0), ("photographic, no burn", 0.0)):
    desc.opt2 = burn
    for _ in range(2):
        assert g.rt_tonemap(d_hdr.data_ptr(), W, H, C.byref(desc), d_out.data_ptr()) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        g.rt_tonemap(d_hdr.data_ptr(), W, H, C.byref(desc), d_out.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    print(f"gpu  {label}: {ms:.3f} ms/frame, {W * H * 15 / ms / 1e6:.1f} GB/s of the 15 B/pixel minimum (12 B in, 3 B out)")
out = np.empty(hdr.shape, dtype=np.uint8)
t0 = time.perf_counter()
R.host_lib().rth_tonemap(0, 0.18, 1.0, 1.0, 2.2, W, H, hdr.ctypes.data, out.ctypes.data)
print(f"host photographic, burn 1 %: {(time.perf_counter() - t0) * 1e3:.1f} ms/frame (1 core)")
d = np.abs(out.astype(np.int32) - d_out.cpu().numpy().astype(np.int32)) if burn == 0.0 else None
