#!/usr/bin/env python3
"""GPU box: how much of a resident Z-slab extraction of the C driver is the ONE host thread enqueueing every rank's launches.
A thin volume (little device work) cut into 1, 2, 4 and 8 slabs on one device: the device work grows slowly with the slab count
(recomputed halos), the launches grow with it; what the wall time does says which one an eight-GPU run would wait for.
usage: python tools/zslab_host_cost.py [NX=128] [NY=128] [NZ=512] [reps=10]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nz = int(sys.argv[3]) if len(sys.argv) > 3 else 512
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
vol = pkg.synth_blobs(nx, ny, nz, seed=12345)
with pkg.Context(nx, ny, nz) as ctx:
    ctx.set_volume(vol)
    want = ctx.extract()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.extract(copy=False)
    print("single context: %.3f ms per extraction, %d records" % ((time.perf_counter() - t0) / reps * 1e3, len(want)))
for s in (1, 2, 4, 8):
    with pkg.ZSlab(nx, ny, nz, [0] * s) as h:
        h.set_volume(vol)
        got, st = h.extract_resident()
        got, st = h.extract_resident(copy=False)
        same = got.tobytes() == want.tobytes()
        t0 = time.perf_counter()
        for _ in range(reps):
            got, st = h.extract_resident(copy=False)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("%d slab(s) on one device: %.3f ms per extraction, of which the host spent %.3f ms enqueueing the pyramids of all ranks (ranks %d, sharded octaves %d, exchanges %d, merge %.2f ms), same bytes %s"
              % (s, ms, st["enqueue_ms"], st["n_ranks"], st["sharded_octaves"], st["exchanges"], st["merge_ms"], same), flush=True)
