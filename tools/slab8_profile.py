import importlib, sys, time
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("cuda-flow3d_amd")
n, r = 512, 8
f0, f1 = pkg.synth_pair(n, n, n)
flow = pkg.SlabOpticalFlow(r, list(range(r)), halo_capacity=32)
flow.initialize(n, n, n)
flow.upload(f0, f1)
t = flow.compute_resident()
t = flow.compute_resident()
print("8 virtual ranks", t)
flow.destroy()
