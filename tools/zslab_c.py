#!/usr/bin/env python3
"""Development aid (GPU box): sift3d_extract_zslab -- the C-level, single-process Z-slab driver -- on a volume cut into one
slab per listed device (repeat a device to rehearse on one GPU), timed against the single-device extraction of the same
volume, with the records compared byte for byte.
usage: python tools/zslab_c.py NX NY NZ dev[,dev...] [reps=3]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
nx, ny, nz = (int(v) for v in sys.argv[1:4])
devs = [int(v) for v in sys.argv[4].split(",")]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
vol = pkg.synth_blobs(nx, ny, nz, seed=12345)
for it in range(reps):
    t0 = time.perf_counter(); recs, st = pkg.extract_zslab(vol, devs); dt = time.perf_counter() - t0
    print("z-slabs over devices %s: %d records in %.1f ms wall (incl. contexts, upload of %.2f GB, download); %s" % (devs, len(recs), dt * 1e3, vol.nbytes / 1e9, st), flush=True)
with pkg.Context(nx, ny, nz, device=devs[0]) as ctx:
    t0 = time.perf_counter(); ctx.set_volume(vol); want = ctx.extract(); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx.extract(copy=False); dt2 = time.perf_counter() - t0
print("single device: %d records, %.1f ms with upload, %.1f ms resident; identical bytes: %s" % (len(want), dt * 1e3, dt2 * 1e3, recs.tobytes() == want.tobytes()))
