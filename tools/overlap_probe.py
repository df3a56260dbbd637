#!/usr/bin/env python3
"""Development aid (GPU box, development build: make -C 3d_sift_cuda_amd/csrc DEV=1): do the keypoint kernel (LDS atomics)
and the descriptor kernel (L1 / TA) gain from sharing the CUs?  sift3d_dev_overlap_probe runs both again on the data of the
last extraction: one after the other, and in alternating slices on two streams.
usage: python tools/overlap_probe.py [N=512] ["kslice:dslice ..."]"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
pkg = importlib.import_module("3d_sift_cuda_amd")
pkg.LIB_HIP = os.path.join(pkg.CSRC, "_build_dev", "libsift3d_hip.so")
L = pkg.hip_lib()
L.sift3d_dev_overlap_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
L.sift3d_dev_overlap_probe.restype = ctypes.c_int
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pairs = (sys.argv[2] if len(sys.argv) > 2 else "448:896 896:1792 1792:3584 3584:7168 448:3584 7168:28672").split()
ctx = pkg.Context(n, n, n)
ctx.set_volume(pkg.synth_blobs(n, n, n))
ctx.set_tuning(pkg.TUNE_SPLIT_TAIL, 0)
ctx.extract(); f = ctx.extract()
print("%d records" % len(f))
out = (ctypes.c_double * 2)()
for pr in pairs:
    k, d = (int(v) for v in pr.split(":"))
    for rep in range(2):
        rc = L.sift3d_dev_overlap_probe(ctx.handle, k, d, out)
        assert rc == 0, rc
        print("slices %5d extrema / %5d records: one after the other %.3f ms, alternating slices on two streams %.3f ms" % (k, d, out[0], out[1]), flush=True)
