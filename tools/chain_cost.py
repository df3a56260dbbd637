#!/usr/bin/env python3
"""What the coarse octaves cost the step: one 512^3 extraction with every octave against the same with the pyramid stopped
after 1, 2, 3 ... octaves (sift3d_set_max_octaves).  The octaves from the third on hold 2 % of the candidates, so what the
step loses when they are cut is what their chain of small launches adds to the critical path.
usage: python tools/chain_cost.py [N=512] [steps=20]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
vol = pkg.synth_blobs(n, n, n, seed=12345)
ctx = pkg.Context(n, n, n)
d = torch.from_numpy(vol).cuda()
torch.cuda.synchronize()
ctx.set_volume_dev(d.data_ptr(), n, n, n)
ctx.sync()
for rep in range(2):
    for mo in (0, 1, 2, 3, 4, 5, 0):
        ctx.set_max_octaves(mo)
        for _ in range(3):
            f = ctx.extract(copy=False)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            f = ctx.extract(copy=False)
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        t = ctx.timings()
        print("max_octaves %d: %.3f ms per step, %d records, %d extrema, %d octaves" % (mo, ms, len(f), t["n_extrema"], t["n_octaves"]), flush=True)
