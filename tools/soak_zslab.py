#!/usr/bin/env python3
"""GPU box: the C slab driver's resident extraction many times through one handle, with the transport (peer copies / the RCCL branch
through the rehearsal library), the communicator sets, the schedule and the level form changed from run to run; every result must be
the single-GPU bytes.  usage: python tools/soak_zslab.py [NX=128] [NY=96] [NZ=512] [ranks=4] [runs=400] [seed=1]"""
import hashlib, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_sift_cuda_amd")
a = [int(v) for v in sys.argv[1:]] + [None] * 6
nx, ny, nz, ranks, runs, seed = a[0] or 128, a[1] or 96, a[2] or 512, a[3] or 4, a[4] or 400, a[5] or 1
rng = np.random.default_rng(seed)
vol = pkg.synth_blobs(nx, ny, nz, seed=77)
with pkg.Context(nx, ny, nz) as ctx:
    ctx.set_volume(vol)
    want = {m: hashlib.sha256(ctx.extract(desc_mode=m).tobytes()).hexdigest() for m in range(4)}
pkg.zslab_set_transport_library(os.path.join(ROOT, "tests", "rccl_shim", "_build", "librccl_shim.so"))
bad, t0 = 0, time.time()
with pkg.ZSlab(nx, ny, nz, [0] * ranks) as h:
    h.set_tuning(pkg.ZSLAB_DUPLICATE_RANKS, 1)
    h.set_volume(vol)
    for i in range(runs):
        mode = int(rng.integers(0, 4))
        if i % 11 == 0:   # (a change of transport rebuilds it: not every run)
            h.set_tuning(pkg.ZSLAB_TRANSPORT, int(rng.choice([pkg.TRANSPORT_PEER_COPY, pkg.TRANSPORT_RCCL])))
            h.set_tuning(pkg.ZSLAB_SERIAL_CHANNELS, int(rng.integers(0, 2)))
        h.set_tuning(pkg.TUNE_BANDS_FIRST, int(rng.choice([1, 1, 0])))
        h.set_tuning(pkg.TUNE_LAZY_LEVELS, int(rng.choice([1, 1, 0])))
        h.set_tuning(pkg.ZSLAB_PATCH_WAIT, int(rng.choice([0, 0, 1])))
        h.set_tuning(pkg.ZSLAB_POISON_HALO, int(rng.choice([0, 1])))
        recs, st = h.extract_resident(desc_mode=mode, copy=False)
        if hashlib.sha256(recs.tobytes()).hexdigest() != want[mode]:
            bad += 1
            print("run %d mode %d transport %d: different bytes" % (i, mode, st["transport"]), flush=True)
        if i % 100 == 0:
            print("run %d (%.0f s)" % (i, time.time() - t0), flush=True)
print("%d resident Z-slab extractions of %d x %d x %d in %d slabs, %d differing, %.0f s" % (runs, nx, ny, nz, ranks, bad, time.time() - t0))
sys.exit(1 if bad else 0)
