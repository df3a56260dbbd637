#!/usr/bin/env python3
"""Development aid (GPU box): time the keypoint kernel cut after stage N.  Needs the development build of the library
(`make -C 3d_sift_cuda_amd/csrc DEV=1` -> csrc/_build_dev), whose kernels carry the ablation branches and which exports
sift3d_dev_set_stop; the product library has neither.  usage: [ABL_N=256] [ABL_STOPS=...] python tools/kp_ablate.py"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
pkg.LIB_HIP = os.path.join(pkg.CSRC, "_build_dev", "libsift3d_hip.so")
L = pkg.hip_lib()
L.sift3d_dev_set_stop.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = int(os.environ.get("ABL_N", "256"))
ctx = pkg.Context(n, n, n)
ctx.set_volume(pkg.synth_blobs(n, n, n))
ctx.set_tuning(pkg.TUNE_KP_CHUNKS, 1)
for stop in [int(v) for v in os.environ.get("ABL_STOPS", "1,2,3,4,7,8,31,32,33,34,0").split(",")]:
    L.sift3d_dev_set_stop(ctx.handle, stop)
    ctx.extract(); ctx.enable_timing(1); ctx.extract()
    log = ctx.launch_log(); sel = log[log["stage"] == 5]
    ctx.enable_timing(0)
    print("stop=%d keypoint ms %.3f (items %s)" % (stop, sel["ms"].sum(), sel["nvox"].tolist()))
