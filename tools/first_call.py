#!/usr/bin/env python3
"""GPU box: what the FIRST extraction of a process costs beyond a warm one (the command line only ever runs the first).
    python tools/first_call.py [n=512]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = pkg.synth_blobs(n, n, n, seed=12345)
t0 = time.perf_counter(); nd = pkg.device_count(); t1 = time.perf_counter()
print("first HIP call (sift3d_device_count): %.3f s" % (t1 - t0))
t0 = time.perf_counter(); ctx = pkg.Context(n, n, n); t1 = time.perf_counter()
print("sift3d_create(%d^3): %.3f s" % (n, t1 - t0))
t0 = time.perf_counter(); ctx.set_volume(vol); t1 = time.perf_counter()
print("sift3d_set_volume (pageable host memory): %.3f s" % (t1 - t0))
for i in range(4):
    t0 = time.perf_counter(); f = ctx.extract(copy=True); t1 = time.perf_counter()
    print("sift3d_extract #%d: %.4f s (%d records)" % (i, t1 - t0, len(f)))
for i in range(2):
    t0 = time.perf_counter(); f = ctx.extract(copy=False); t1 = time.perf_counter()
    print("sift3d_extract_view #%d: %.4f s" % (i, t1 - t0))
t0 = time.perf_counter(); ctx.close(); t1 = time.perf_counter()
print("sift3d_destroy: %.3f s" % (t1 - t0))
