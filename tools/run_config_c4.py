#!/usr/bin/env python3
"""BASELINE config C4's volume (1024 x 1024 x 512 float32) on ONE GPU: timing and sanity checks (the 4-GPU Z-slab run of
the same volume is what `bench.py --gpus 4 --size ...` / zslab.py do; this shows the single-GPU cost it is compared with)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
nx, ny, nz = 1024, 1024, 512
t0 = time.time(); vol = pkg.synth_blobs(nx, ny, nz); print("synth %.1fs" % (time.time() - t0), flush=True)
ctx = pkg.Context(nx, ny, nz)
d = torch.from_numpy(vol).cuda(); torch.cuda.synchronize()
ctx.set_volume_dev(d.data_ptr(), nx, ny, nz); ctx.sync()
for it in range(3):
    t0 = time.time(); f = ctx.extract(copy=False); dt = time.time() - t0
    tim = ctx.timings()
    print("run %d: %d records, %d extrema, %d keypoints, %d octaves, wall %.1f ms" % (it, len(f), tim["n_extrema"], tim["n_keypoints"], tim["n_octaves"], dt * 1e3), flush=True)
f = f.copy()
assert np.isfinite(f["x"]).all() and (f["x"] <= nx).all() and (f["y"] <= ny).all() and (f["z"] <= nz).all()
again = ctx.extract()
assert (again.view(np.uint8) == f.view(np.uint8)).all()
print("C4 volume ok: records per voxel 1/%.0f" % (nx * ny * nz / len(f)))
