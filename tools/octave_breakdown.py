#!/usr/bin/env python3
"""Per-octave / per-stage device time of one extraction (HIP-event launch log).
usage: python tools/octave_breakdown.py [N=512]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = pkg.synth_blobs(n, n, n)
ctx = pkg.Context(n, n, n)
ctx.set_volume(vol)
ctx.extract(); ctx.extract()
ctx.enable_timing(True)
t0 = time.perf_counter(); f = ctx.extract(); wall = time.perf_counter() - t0
log = ctx.launch_log(); tim = ctx.timings()
print("records %d  wall %.2f ms  stream first->last %.2f ms  sum of kernels %.2f ms" % (len(f), wall * 1e3, tim["total_ms"], log["ms"].sum()))
ctx.enable_timing(False)
t0 = time.perf_counter(); f = ctx.extract(); wall2 = time.perf_counter() - t0
print("wall without event timing %.2f ms" % (wall2 * 1e3))
vol_stages = (0, 1, 2, 7, 3, 4)
sizes = sorted(set(int(v) for v in log["nvox"][np.isin(log["stage"], vol_stages)]), reverse=True)
print("%-12s" % "nvox" + "".join("%12s" % pkg.STAGES[s] for s in vol_stages))
for nv in sizes:
    row = []
    for st in vol_stages:
        sel = log[(log["stage"] == st) & (log["nvox"] == nv)]
        row.append("%7.3f(%2d)" % (sel["ms"].sum(), len(sel)))
    print("%-12d" % nv + "".join("%12s" % r for r in row))
for st in (5, 6):
    sel = log[log["stage"] == st]
    print(pkg.STAGES[st], "launches", len(sel), "items", sel["nvox"].tolist(), "ms", np.round(sel["ms"], 3).tolist())
sel = log[log["stage"] == 7]
print("fused launches (voxels, taps, ms):", [(int(r["nvox"]), int(r["ntaps"]), round(float(r["ms"]), 3)) for r in sel])
