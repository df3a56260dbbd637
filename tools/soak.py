#!/usr/bin/env python3
"""GPU box: the same extraction many times on one context, with the schedule knobs changed from run to run; every result
must be the same bytes (a race between the streams of the two-part schedule would show as a difference).
usage: python tools/soak.py [N=256] [runs=2000] [seed=1]"""
import hashlib, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
vol = pkg.synth_blobs(n, n, n, seed=4242)
bad = 0
t0 = time.time()
with pkg.Context(n, n, n) as ctx:
    ctx.set_volume(vol)
    want = {}
    for i in range(runs):
        mode = int(rng.integers(0, 4))
        ctx.set_tuning(pkg.TUNE_SPLIT_TAIL, int(rng.choice([1, 1, 1, 0, 2])))
        ctx.set_tuning(pkg.TUNE_FUSED_SUB, int(rng.choice([1, 1, 0])))
        ctx.set_tuning(pkg.TUNE_DESC_SEGMENT, int(rng.choice([32, 32, 0, 1, 7])))
        ctx.set_tuning(pkg.TUNE_KP_CHUNKS, int(rng.choice([0, 0, 0, 3])))
        ctx.set_tuning(pkg.TUNE_FUSED_ORDER, int(rng.choice([0, 0, 1, 2, 3])))   # round 5
        ctx.set_tuning(pkg.TUNE_FUSED_STAGGER, int(rng.choice([0, 0, 1, 2])))    # round 6
        if i % 97 == 5:   # round 5: the volume handed over again, in runs of planes
            cut = int(rng.integers(1, n))
            ctx.set_volume_in_runs(vol, [(cut, n - cut), (0, cut)])
        h = hashlib.sha256(ctx.extract(desc_mode=mode, copy=False).tobytes()).hexdigest()
        if mode not in want:
            want[mode] = h
        elif want[mode] != h:
            bad += 1
            print("run %d mode %d: different bytes" % (i, mode), flush=True)
        if i % 500 == 0:
            print("run %d (%.0f s)" % (i, time.time() - t0), flush=True)
print("%d runs at %d^3, %d differing, %.0f s" % (runs, n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
