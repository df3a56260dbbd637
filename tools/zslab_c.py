#!/usr/bin/env python3
"""Development aid (GPU box): sift3d_extract_zslab -- the C-level, single-process Z-slab driver -- on a volume cut into one
slab per listed device (repeat a device to rehearse on one GPU), timed against the single-device extraction of the same
volume, with the records compared byte for byte.
usage: python tools/zslab_c.py NX NY NZ dev[,dev...] [reps=3] [bands_first=1]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
nx, ny, nz = (int(v) for v in sys.argv[1:4])
devs = [int(v) for v in sys.argv[4].split(",")]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
bands = int(sys.argv[6]) if len(sys.argv) > 6 else 1
vol = pkg.synth_blobs(nx, ny, nz, seed=12345)
t0 = time.perf_counter(); recs, st = pkg.extract_zslab(vol, devs); dt = time.perf_counter() - t0
print("one-shot (contexts created and destroyed inside the call): %d records in %.1f ms wall; %s" % (len(recs), dt * 1e3, st), flush=True)
t0 = time.perf_counter()
with pkg.ZSlab(nx, ny, nz, devs) as h:
    print("handle created in %.1f ms; boundary bands first: %d" % ((time.perf_counter() - t0) * 1e3, bands), flush=True)
    h.set_tuning(pkg.TUNE_BANDS_FIRST, bands)
    for it in range(reps + 1):
        recs, st = h.extract(vol)
        print("  extract %d through the handle: %d records, %.1f ms wall (upload of %.2f GB of slabs, pyramid, per-keypoint stage, download, merge)%s"
              % (it, len(recs), st["wall_ms"], vol.nbytes / 1e9, "  [first run: buffers allocated one by one, arena sized afterwards]" if it == 0 else ""), flush=True)
with pkg.Context(nx, ny, nz, device=devs[0]) as ctx:
    t0 = time.perf_counter(); ctx.set_volume(vol); want = ctx.extract(); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx.set_volume(vol); ctx.extract(); dt1 = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx.extract(copy=False); dt2 = time.perf_counter() - t0
print("single device: %d records, %.1f ms with upload (first call %.1f), %.1f ms resident; identical bytes: %s" % (len(want), dt1 * 1e3, dt * 1e3, dt2 * 1e3, recs.tobytes() == want.tobytes()))
