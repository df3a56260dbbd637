#!/usr/bin/env python3
"""Development aid (GPU box; needs `make -C 3d_sift_cuda_amd/csrc DEV=1`): the keypoint kernel with the SVD + eigen test BEFORE the
first orientation splat (stop code 40) against the product's order (the SVD on one lane BESIDE the splat), alternated.
Round-4 review item 5b: 16 % of a blob field's extrema fail the eigen test after their splat has run.
usage: python tools/kp_svd_first.py [n=512] [pairs=4]"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
pkg.LIB_HIP = os.path.join(pkg.CSRC, "_build_dev", "libsift3d_hip.so")
L = pkg.hip_lib()
L.sift3d_dev_set_stop.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = pkg.Context(n, n, n)
ctx.set_volume(pkg.synth_blobs(n, n, n))
ref = None
for pair in range(pairs):
    for stop, name in ((0, "SVD beside the splat (product)"), (40, "SVD before the splat")):
        L.sift3d_dev_set_stop(ctx.handle, stop)
        ctx.extract(copy=False)
        t0 = time.perf_counter()
        for _ in range(10):
            f = ctx.extract(copy=False)
        step = (time.perf_counter() - t0) / 10 * 1e3
        b = f.tobytes()
        if ref is None:
            ref = b
        ctx.set_tuning(pkg.TUNE_SPLIT_TAIL, 0); ctx.enable_timing(1); ctx.extract(copy=False)
        log = ctx.launch_log(); kp = log[log["stage"] == 5]["ms"].sum(); ds = log[log["stage"] == 6]["ms"].sum()
        ctx.enable_timing(0); ctx.set_tuning(pkg.TUNE_SPLIT_TAIL, 1)
        print("pair %d  %-32s step %.3f ms   keypoint kernel %.3f ms   descriptor kernel %.3f ms   records %d  same bytes %s"
              % (pair, name, step, kp, ds, len(f), b == ref), flush=True)
