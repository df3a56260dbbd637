#!/usr/bin/env python3
"""GPU box: does the PLACEMENT of a blur launch's three buffers matter to the fused kernel as it does to the zero-arithmetic march
(tools/roof_placement.py)?  Input, level and DoG as three 2^29-byte buffers back to back, or with the two outputs shifted by odd
amounts; per instantiation the median launch time.  usage: python tools/ab_skew.py [reps=15] [rounds=3]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 512; N = n ** 3
ctx = pkg.Context(n, n, n)
big = torch.empty(3 * N + (8 << 20), device="cuda")
big[:N] = torch.randn(N, device="cuda") * 50
torch.cuda.synchronize()
sig = {7: 1.2262736558914185, 9: 1.5450079441070557, 11: 1.9465880393981934, 13: 2.452547311782837}
skews = [0, 271552, (3 << 20) + (37 << 10) + 448, 64, 1024, 16384]
print("fused blur at 512^3, input / level / DoG = three 2^29-byte buffers; median of %d launches, ms" % reps)
print("%-22s " % "outputs shifted by" + " ".join("%12s" % ("%d taps+DoG" % t) for t in sig) + " %12s %12s" % ("7 level", "9 level"))
for r in range(rounds):
    for s in skews:
        a, b, d = big[:N], big[N + s:2 * N + s], big[2 * N + 2 * s:3 * N + 2 * s]
        row = []
        for taps, sg in list(sig.items()) + [(-7, sig[7]), (-9, 1.5198684930801392)]:
            def launch():
                if taps > 0:
                    ctx.gauss_blur_dog_dev(a.data_ptr(), b.data_ptr(), d.data_ptr(), n, n, n, sg)
                else:
                    ctx.gauss_blur_dev(a.data_ptr(), b.data_ptr(), n, n, n, sg)
            for _ in range(2):
                launch()
            ctx.enable_timing(True)
            for _ in range(reps):
                launch()
            log = ctx.launch_log()
            ctx.enable_timing(False)
            row.append(float(np.median(log[log["stage"] == 7]["ms"])))
        print("%-22s " % ("%d floats (%.2f MB)" % (s, s * 4 / 1e6)) + " ".join("%12.4f" % v for v in row), flush=True)
