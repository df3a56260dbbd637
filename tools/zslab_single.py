#!/usr/bin/env python3
"""GPU box: the Z-slab driver with one rank (no exchange) against the fused pipeline: what the operator-level,
Python-driven path costs before any communication.  usage: python tools/zslab_single.py [N=512] [steps=5]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
zs = importlib.import_module("3d_sift_cuda_amd.zslab")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
vol = pkg.synth_blobs(n, n, n, seed=12345)
dvol = torch.from_numpy(vol).cuda(); torch.cuda.synchronize()
plan = zs.SlabPlan(n, n, n, 1)
ctx = pkg.Context(n, n, n + 2 * zs.HALO, slab=True)
be = zs.HipBackend(pkg, ctx, torch)


def step():
    with be.stream_scope():
        ex = zs.ZSlabExtractor(be, plan, 0, None)
        t0 = time.perf_counter(); ex.run(dvol, 0); torch.cuda.synchronize(); t1 = time.perf_counter()
        recs, grp = ex.describe(desc_mode=0, copy=False); t2 = time.perf_counter()
    return recs, t1 - t0, t2 - t1


for _ in range(2):
    step()
tr = td = 0.0
for _ in range(steps):
    recs, a, b = step(); tr += a; td += b
print("zslab driver, 1 rank, %d^3: run (upload + pyramid + extrema) %.2f ms, describe %.2f ms, %d records" % (n, 1e3 * tr / steps, 1e3 * td / steps, len(recs)))
ctx2 = pkg.Context(n, n, n); ctx2.set_volume(vol)
for _ in range(2):
    ctx2.extract(copy=False)
t0 = time.perf_counter()
for _ in range(steps):
    f = ctx2.extract(copy=False)
print("fused pipeline: %.2f ms, %d records" % (1e3 * (time.perf_counter() - t0) / steps, len(f)))
