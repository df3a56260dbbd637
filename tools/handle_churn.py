#!/usr/bin/env python3
"""GPU box: forty slab handles (2 - 5 ranks on one device) created, used once and destroyed: the extraction's bytes stay the same,
the process's thread count and peak memory stay where they were after the first ten (the handle's host threads are joined, its
pinned list and arenas freed).  usage: python tools/handle_churn.py"""
import importlib, os, sys, threading, resource
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("3d_sift_cuda_amd")
vol = pkg.synth_blobs(96, 80, 256, seed=3)
def nthreads():
    return len(os.listdir("/proc/self/task"))
want = None
for i in range(40):
    with pkg.ZSlab(96, 80, 256, [0] * (2 + i % 4)) as h:
        got, st = h.extract(vol)
        if want is None: want = got.tobytes()
        assert got.tobytes() == want
    if i % 10 == 9:
        print(i, "threads", nthreads(), "maxrss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, flush=True)
