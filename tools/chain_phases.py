#!/usr/bin/env python3
"""Development aid (GPU box): where the coarse-octave launch (kernels_chain.hip) spends its time.  Needs the development
build (`make -C 3d_sift_cuda_amd/csrc DEV=1`), whose kernel leaves workgroup 0's 100 MHz clock before and after every grid
barrier.  usage: python tools/chain_phases.py [N=512] [WGS=48]"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
pkg.LIB_HIP = os.path.join(pkg.CSRC, "_build_dev", "libsift3d_hip.so")
L = pkg.hip_lib()
L.sift3d_dev_chain_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 48
ctx = pkg.Context(n, n, n)
ctx.set_volume(pkg.synth_blobs(n, n, n))
ctx.set_tuning(pkg.TUNE_COARSE_CHAIN, wgs)
buf = np.zeros(256, np.uint64)
for mode in (3, 0):   # 3: the chain alone on the main stream; 0: the production schedule (beside the extrema passes)
    ctx.enable_timing(mode)
    ctx.extract(); ctx.extract()
    L.sift3d_dev_chain_clocks(buf.ctypes.data, 256)
    ctx.extract()
    k = L.sift3d_dev_chain_clocks(buf.ctypes.data, 256)
    t = (buf[:k] - buf[0]).astype(np.float64) / 100.0   # microseconds
    print("timing mode %d: %d stamps, total %.1f us" % (mode, k, t[-1]))
    # stamps: start, then (work end, barrier end) per grid phase, then solo phases
    work = t[1::2][:45] - t[0::2][:45]
    bar = t[2::2][:45] - t[1::2][:45]
    for o in range(3):
        w, b = work[15 * o:15 * o + 15], bar[15 * o:15 * o + 15]
        print("  octave %d: work %s" % (o, " ".join("%5.1f" % v for v in w)))
        print("            wait %s" % " ".join("%5.1f" % v for v in b))
        print("            sum work %.1f, barriers %.1f" % (w.sum(), b.sum()))
    print("  after the grid phases (workgroup 0 alone): %.1f us" % (t[-1] - t[90] if k > 91 else 0.0))
