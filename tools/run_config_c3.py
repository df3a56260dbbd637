#!/usr/bin/env python3
"""BASELINE config C3: 512^3 float32 volume, -2+ upsample (processing size 1024^3), BRIEF descriptor, 1 GPU.
Prints timing and size-independent sanity checks (the oracle cannot finish 1024^3 in test time)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t0 = time.time(); vol = pkg.synth_blobs(n, n, n); print("synth %.1fs" % (time.time() - t0), flush=True)
ctx = pkg.Context(2 * n, 2 * n, 2 * n)
t0 = time.time(); big = ctx.double_size(vol); print("double_size (incl. H2D/D2H of %.1f GB) %.2fs" % (big.nbytes / 1e9, time.time() - t0), flush=True)
ctx.set_volume(big)
ctx.enable_timing(True)
for it in range(2):
    t0 = time.time(); f = ctx.extract(initial_image_scale=0.5, desc_mode=pkg.DESC_BRIEF, size_factor=0.5, copy=False); dt = time.time() - t0
    tim = ctx.timings()
    print("run %d: %d records, %d extrema, %d octaves, wall %.1f ms, stream %.1f ms, stages %s" % (
        it, len(f), tim["n_extrema"], tim["n_octaves"], dt * 1e3, tim["total_ms"],
        {k: round(v["ms"], 2) for k, v in tim["stages"].items()}), flush=True)
f = f.copy()
assert np.isfinite(f["x"]).all() and (f["x"] >= 0).all() and (f["x"] <= n).all() and (f["z"] <= n).all()
assert set(np.unique(f["desc"])) <= set(np.arange(64.0))          # rank descriptors
assert ((f["info"] & ~np.uint32(0x30)) == 0).all()
again = ctx.extract(initial_image_scale=0.5, desc_mode=pkg.DESC_BRIEF, size_factor=0.5)
assert (again.view(np.uint8) == f.view(np.uint8)).all()
print("C3 ok: records per input voxel 1/%.0f" % (n ** 3 / len(f)))
