#!/usr/bin/env python3
"""GPU box: the Z-slab driver with one rank (no exchange) against the fused pipeline: what the operator-level,
Python-driven path costs before any communication -- and how long the host takes to QUEUE a rank's pyramid (run() returning,
nothing waited for), which at eight ranks is most of a step.  usage: python tools/zslab_single.py [N=512] [steps=5] [NZ=N]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
zs = importlib.import_module("3d_sift_cuda_amd.zslab")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nz = int(sys.argv[3]) if len(sys.argv) > 3 else n
vol = pkg.synth_blobs(n, n, nz, seed=12345)
dvol = torch.from_numpy(vol).cuda(); torch.cuda.synchronize()
plan = zs.SlabPlan(n, n, nz, 1)
ctx = pkg.Context(n, n, nz + 2 * zs.HALO, slab=True)
be = zs.HipBackend(pkg, ctx, torch)


def step():
    with be.stream_scope():
        ex = zs.ZSlabExtractor(be, plan, 0, None)
        t0 = time.perf_counter(); ex.run(dvol, 0); tq = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
        recs, grp = ex.describe(desc_mode=0, copy=False); t2 = time.perf_counter()
    return recs, t1 - t0, t2 - t1, tq - t0


for _ in range(2):
    step()
tr = td = tq = 0.0
for _ in range(steps):
    recs, a, b, c = step(); tr += a; td += b; tq += c
print("zslab driver, 1 rank, %d x %d x %d: run (pyramid + extrema) %.2f ms of which the host needed %.2f ms to queue it, describe %.2f ms, %d records"
      % (n, n, nz, 1e3 * tr / steps, 1e3 * tq / steps, 1e3 * td / steps, len(recs)))
ctx2 = pkg.Context(n, n, nz); ctx2.set_volume(vol)
for _ in range(2):
    ctx2.extract(copy=False)
t0 = time.perf_counter()
for _ in range(steps):
    f = ctx2.extract(copy=False)
print("fused pipeline: %.2f ms, %d records" % (1e3 * (time.perf_counter() - t0) / steps, len(f)))
