#!/usr/bin/env python3
"""GPU box: throughput of K contexts extracting independent 512^3 volumes concurrently on ONE GPU (one host thread and
one set of streams per context), against one context.  The bench line measures one volume at a time; this is what a
serving loop that keeps several volumes in flight gets.  usage: python tools/two_volumes.py [N=512] [K=2] [reps=10]"""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctxs = []
for k in range(K):
    c = pkg.Context(n, n, n)
    c.set_volume(pkg.synth_blobs(n, n, n, seed=12345 + k))
    c.extract(copy=False); c.extract(copy=False)
    ctxs.append(c)
t0 = time.perf_counter()
for _ in range(reps):
    nrec = len(ctxs[0].extract(copy=False))
one = (time.perf_counter() - t0) / reps
print("one context: %.2f ms per volume (%d records)" % (1e3 * one, nrec))
counts = [0] * K


def worker(k):
    for _ in range(reps):
        counts[k] = len(ctxs[k].extract(copy=False))


th = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
print("%d contexts in flight: %.2f ms per volume (%.2f ms per round of %d), records %s" % (K, 1e3 * dt / (reps * K), 1e3 * dt / reps, K, counts))
