#!/usr/bin/env python3
"""Per-kernel timing of the blur passes at one volume size (HIP events via the C-ABI launch log).
usage: python tools/bench_blur.py [N=512] [reps=10] [tile=0] [order=0] [stagger=0]   (stagger: SIFT3D_TUNE_FUSED_STAGGER, 1 = off, the
kernel of rounds 2 - 5, 2 = on: one copy of the march per wavefront role and the half-step stagger; tile: SIFT3D_TUNE_FUSED_TILE, 1 = 64 x 32, 2 = 128 x 16;
order: SIFT3D_TUNE_FUSED_ORDER, 1 = rows of tiles per XCD, 2 = workgroup b takes tile b, 3 = column strips per XCD)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
if os.environ.get("SIFT3D_AB_BUILD"):   # A/B of two builds of the library on one box: csrc/<dir>/libsift3d_hip.so instead of csrc/_build
    pkg.LIB_HIP = os.path.join(pkg.CSRC, os.environ["SIFT3D_AB_BUILD"], "libsift3d_hip.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 0
order = int(sys.argv[4]) if len(sys.argv) > 4 else 0
stagger = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx = pkg.Context(n, n, n)
ctx.set_tuning(pkg.TUNE_FUSED_TILE, tile)
ctx.set_tuning(pkg.TUNE_FUSED_ORDER, order)
ctx.set_tuning(pkg.TUNE_FUSED_STAGGER, stagger)
a = torch.randn(n, n, n, device="cuda") * 50
b = torch.empty_like(a); d = torch.empty_like(a)
torch.cuda.synchronize()
sig = {7: 1.2262736558914185, 9: 1.5450079441070557, 11: 1.9465880393981934, 13: 2.452547311782837, 17: 3.0900158882141113}
N = n ** 3
print("N=%d^3  bytes/pass: %.3f GB  tile knob %d  order knob %d  stagger knob %d" % (n, 8 * N / 1e9, tile, order, stagger))
cases = [(taps, s, True) for taps, s in sig.items()] + [(7, sig[7], False), (9, 1.5198684930801392, False)]   # level-only: L1 and the initial blur
for taps, s, with_dog in cases:
    outp = 0 if taps == 17 else b.data_ptr()   # the 17-tap level, when it is filtered in full (SIFT3D_TUNE_LAZY_LEVELS = 0), is not stored, only its DoG
    if not with_dog:
        for _ in range(2):
            ctx.gauss_blur_dev(a.data_ptr(), b.data_ptr(), n, n, n, s)
        ctx.enable_timing(True)
        for _ in range(reps):
            ctx.gauss_blur_dev(a.data_ptr(), b.data_ptr(), n, n, n, s)
        log = ctx.launch_log()
        ctx.enable_timing(False)
        sel = log[log["stage"] == 7]
        if len(sel):
            ms = float(np.median(sel["ms"]))
            print("taps %2d: fused %.3f ms  %.0f GB/s (8N, level only)" % (taps, ms, 8 * N / ms / 1e6))
        continue
    for _ in range(2):
        ctx.gauss_blur_dog_dev(a.data_ptr(), outp, d.data_ptr(), n, n, n, s)
    ctx.enable_timing(True)
    for _ in range(reps):
        ctx.gauss_blur_dog_dev(a.data_ptr(), outp, d.data_ptr(), n, n, n, s)
    log = ctx.launch_log()
    ctx.enable_timing(False)
    if (log["stage"] == 7).any():
        sel = log[log["stage"] == 7]
        ms = float(np.median(sel["ms"]))
        bpv = 8 if taps == 17 else 12
        print("taps %2d: fused %.3f ms  %.0f GB/s (%dN)" % (taps, ms, bpv * N / ms / 1e6, bpv))
        continue
    out = []
    for st, name in ((0, "x"), (1, "y"), (2, "z+dog")):
        sel = log[log["stage"] == st]
        ms = float(np.median(sel["ms"])); by = float(sel["alg_bytes"][0])
        out.append("%s %.3f ms %.0f GB/s" % (name, ms, by / ms / 1e6))
    tot = sum(float(np.median(log[log["stage"] == st]["ms"])) for st in range(3))
    print("taps %2d: %s | total %.3f ms  %.0f GB/s (32N)" % (taps, " | ".join(out), tot, 32 * N / tot / 1e6))
# the pyramid's level-3 launch: 11 taps, level + DoG + the next octave's level 0 (its own instantiation of the kernel)
h = torch.empty(n // 2, n // 2, n // 2, device="cuda")
for _ in range(2):
    ctx.gauss_blur_dog_half_dev(a.data_ptr(), b.data_ptr(), d.data_ptr(), h.data_ptr(), n, n, n, sig[11])
ctx.enable_timing(True)
for _ in range(reps):
    one = ctx.gauss_blur_dog_half_dev(a.data_ptr(), b.data_ptr(), d.data_ptr(), h.data_ptr(), n, n, n, sig[11])
log = ctx.launch_log()
ctx.enable_timing(False)
ms = float(np.median(log[log["stage"] == 7]["ms"]))
print("taps 11 + half-size volume (%s): fused %.3f ms  %.0f GB/s (12.5N)" % ("one launch" if one else "TWO launches", ms, 12.5 * N / ms / 1e6))
