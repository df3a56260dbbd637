#!/usr/bin/env python3
"""GPU box: round-4 review item 3, second form, MEASURED: the two level-only launches of a 512^3 volume (initial 9-tap blur V -> L0, 7-tap blur
L0 -> L1) cut into z windows and interleaved, so that L1's launch reads the planes L0's launch has just written while they are still in the
256 MB Infinity Cache -- against the two full launches.  Same bytes either way (checked).  Device time by HIP events around the whole sequence.
usage: python tools/pair_windows.py [n=512] [reps=10]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S0, S1 = 1.5198684930801392, 1.2262736558914185   # the initial blur (9 taps) and the first level's (7 taps)
ctx = pkg.Context(n, n, n)
stream = torch.cuda.Stream()
ctx.set_stream(stream.cuda_stream)
V = torch.from_numpy(pkg.synth_blobs(n, n, n, seed=12345)).cuda()
L0, L1 = torch.empty_like(V), torch.empty_like(V)
ref0, ref1 = torch.empty_like(V), torch.empty_like(V)
torch.cuda.synchronize()


def full():
    ctx.gauss_blur_dev(V.data_ptr(), L0.data_ptr(), n, n, n, S0)
    ctx.gauss_blur_dev(L0.data_ptr(), L1.data_ptr(), n, n, n, S1)


def windows(W):
    R1 = 3
    k = 0
    while k * W < n:
        a, b = k * W, min(n, (k + 1) * W)
        ctx.gauss_blur_dog_window_dev(V.data_ptr(), L0.data_ptr(), 0, n, n, n, a, b, S0)
        a1, b1 = max(0, a - R1), (n if b == n else b - R1)
        if b1 > a1:
            ctx.gauss_blur_dog_window_dev(L0.data_ptr(), L1.data_ptr(), 0, n, n, n, a1, b1, S1)
        k += 1


def timed(fn):
    ms = []
    for r in range(reps + 2):
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); fn(); e1.record(stream)
        e1.synchronize()
        if r >= 2:
            ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), float(np.min(ms))


with torch.cuda.stream(stream):
    full()
torch.cuda.synchronize()
ref0.copy_(L0); ref1.copy_(L1)
med, best = timed(full)
print("two full launches (V -> L0, L0 -> L1):                      median %.3f ms, best %.3f ms" % (med, best))
for W in (256, 128, 64, 32):
    L0.zero_(); L1.zero_(); torch.cuda.synchronize()
    med, best = timed(lambda: windows(W))
    torch.cuda.synchronize()
    same = bool(torch.equal(L0, ref0) and torch.equal(L1, ref1))
    print("interleaved in windows of %3d planes (%2d launches):        median %.3f ms, best %.3f ms   same bytes %s" % (W, 2 * ((n + W - 1) // W), med, best, same))
