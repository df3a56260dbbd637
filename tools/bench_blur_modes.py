#!/usr/bin/env python3
"""The pyramid's fused-blur instantiations at one volume size under several SIFT3D_TUNE_FUSED_STAGGER modes, in ONE process (so that one
rocprofv3 --pmc run sees every variant: the mode is the kernel's last template argument).
usage: python tools/bench_blur_modes.py [N=512] [reps=3] [modes=1,2]   (1 = off, 2 = on; the round-6 measurements in
profiles/r06_stagger_ab.txt were taken with a build that had more modes and a prefetch knob: see its header)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
if os.environ.get("SIFT3D_AB_BUILD"):   # A/B of two builds of the library on one box: csrc/<dir>/libsift3d_hip.so instead of csrc/_build
    pkg.LIB_HIP = os.path.join(pkg.CSRC, os.environ["SIFT3D_AB_BUILD"], "libsift3d_hip.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
modes = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "1,2").split(",")]
ctx = pkg.Context(n, n, n)
a = torch.randn(n, n, n, device="cuda") * 50
b = torch.empty_like(a); d = torch.empty_like(a)
h = torch.empty(n // 2, n // 2, n // 2, device="cuda")
torch.cuda.synchronize()
sig = {7: 1.2262736558914185, 9: 1.5450079441070557, 11: 1.9465880393981934, 13: 2.452547311782837}
N = n ** 3
cases = [("9 level (initial)", lambda: ctx.gauss_blur_dev(a.data_ptr(), b.data_ptr(), n, n, n, 1.5198684930801392), 8),
         ("7 level", lambda: ctx.gauss_blur_dev(a.data_ptr(), b.data_ptr(), n, n, n, sig[7]), 8),
         ("9 level+DoG", lambda: ctx.gauss_blur_dog_dev(a.data_ptr(), b.data_ptr(), d.data_ptr(), n, n, n, sig[9]), 12),
         ("11 level+DoG+half", lambda: ctx.gauss_blur_dog_half_dev(a.data_ptr(), b.data_ptr(), d.data_ptr(), h.data_ptr(), n, n, n, sig[11]), 12.5),
         ("13 level+DoG", lambda: ctx.gauss_blur_dog_dev(a.data_ptr(), b.data_ptr(), d.data_ptr(), n, n, n, sig[13]), 12)]
for mode in modes:
    ctx.set_tuning(pkg.TUNE_FUSED_STAGGER, mode)
    tot = 0.0
    for name, fn, bpv in cases:
        fn(); fn()
        ctx.enable_timing(True)
        for _ in range(reps):
            fn()
        log = ctx.launch_log()
        ctx.enable_timing(False)
        ms = float(np.median(log[log["stage"] == 7]["ms"]))
        tot += ms
        print("mode %d  %-20s %.3f ms  %.0f GB/s" % (mode, name, ms, bpv * N / ms / 1e6))
    print("mode %d  five launches %.3f ms  frac of 8 TB/s %.3f" % (mode, tot, 52.5 * N / tot / 1e6 / 8000))
