#!/usr/bin/env python3
"""A few resident Z-slab extractions of the C driver with S slabs on one device, for a rocprofv3 kernel + memory-copy trace
(tools/zslab_trace.sh): what the halo copies run beside, and what waits for them.
usage: python tools/zslab_trace.py [S=2] [patch_wait=0] [reps=3] [N=512]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
wait = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = int(sys.argv[4]) if len(sys.argv) > 4 else 512
vol = pkg.synth_blobs(n, n, n, seed=12345)
with pkg.ZSlab(n, n, n, [0] * S) as h:
    h.set_tuning(pkg.ZSLAB_PATCH_WAIT, wait)
    h.set_volume(vol)
    h.extract_resident(copy=False)
    h.extract_resident(copy=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        got, st = h.extract_resident(copy=False)
    ms = (time.perf_counter() - t0) / reps * 1e3
    for _ in range(2):   # for the trace: extractions with idle time around them
        time.sleep(0.03)
        h.extract_resident(copy=False)
    time.sleep(0.03)
    print("%d slabs, patch_wait %d: %.3f ms per extraction (enqueue %.3f ms), %d records, deferred %d B of which the subsample's %d B, critical %d B"
          % (S, wait, ms, st["enqueue_ms"], len(got), st["halo_bytes_deferred"], st["halo_bytes_subsample"],
             st["halo_bytes_critical"]), flush=True)
