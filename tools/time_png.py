import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as R
h = R.host_lib()
rng = np.random.default_rng(0)
flat = np.zeros((1080, 1920, 3), dtype=np.uint8); flat[300:700, 500:1400] = 128
noisy = rng.integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
os.makedirs('/tmp/pngt', exist_ok=True)
for name, img in (("flat", flat), ("noisy", noisy)):
    t0 = time.perf_counter(); c0 = time.process_time()
    h.rth_write_png(b'/tmp/pngt/x.png', 1920, 1080, img.ctypes.data, 1)
    print(name, "wall", round(1e3 * (time.perf_counter() - t0), 1), "ms, cpu", round(1e3 * (time.process_time() - c0), 1), "ms")
