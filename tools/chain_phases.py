#!/usr/bin/env python3
"""Development aid (GPU box): where the coarse-octave launch (blur_chain_kernel) spends its time.  Needs the development
build (`make -C 3d_sift_cuda_amd/csrc DEV=1`), whose kernel leaves workgroup 0's 100 MHz clock before and after every grid
barrier and after every single-workgroup octave.  usage: python tools/chain_phases.py [N=512] [WGS=32]"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
pkg.LIB_HIP = os.path.join(pkg.CSRC, "_build_dev", "libsift3d_hip.so")
L = pkg.hip_lib()
L.sift3d_dev_chain_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = pkg.Context(n, n, n)
ctx.set_volume(pkg.synth_blobs(n, n, n))
ctx.set_tuning(pkg.TUNE_BLUR_CHAIN, wgs)
buf = np.zeros(256, np.uint64)
for mode in (3, 0):   # 3: the chain alone on the main stream; 0: the production schedule (beside the extrema passes)
    ctx.enable_timing(mode)
    ctx.extract(); ctx.extract()
    L.sift3d_dev_chain_clocks(buf.ctypes.data, 256)
    ctx.extract()
    k = L.sift3d_dev_chain_clocks(buf.ctypes.data, 256)
    t = (buf[:k] - buf[0]).astype(np.float64) / 100.0   # microseconds
    print("timing mode %d, %d workgroups: %d stamps, total %.1f us" % (mode, wgs, k, t[-1]))
    ngrid = (k - 1 - 3) // 8          # stamps: start, (work end, barrier end) x 4 levels per grid octave, then one per single-workgroup octave
    for o in range(ngrid):
        w = [t[1 + 8 * o + 2 * j] - t[8 * o + 2 * j] for j in range(4)]
        b = [t[2 + 8 * o + 2 * j] - t[1 + 8 * o + 2 * j] for j in range(4)]
        print("  grid octave %d: work %s   barrier %s" % (o, " ".join("%6.1f" % v for v in w), " ".join("%5.1f" % v for v in b)))
    tail = t[1 + 8 * ngrid:] - t[8 * ngrid:-1]
    print("  single-workgroup octaves: %s" % " ".join("%6.1f" % v for v in tail))
