import importlib, sys, ctypes as C, time
sys.path.insert(0, "/root/repo")
import numpy as np
pkg = importlib.import_module("cuda-flow3d_amd"); hip = pkg.hip()
W, H, D = 512, 512, 74
box = pkg.Containers(W, H, D)
rng = np.random.default_rng(0)
vol = rng.uniform(-3, 3, (D, H, W)).astype(np.float32); vol[5, 7, 9] = np.nan; vol[40, 300, 400] = -7.25
p = box.new(vol); box.set_current()
r = C.c_float()
pkg.check(hip.f3d_abs_max(p, W, H, D, None, C.byref(r))); assert r.value == 7.25, r.value
t0 = time.time()
for _ in range(200): pkg.check(hip.f3d_abs_max(p, W, H, D, None, C.byref(r)))
print(f"abs_max {W}x{H}x{D}: {(time.time() - t0) / 200 * 1e6:.1f} us per call incl. readback, value {r.value}")
box.free()
