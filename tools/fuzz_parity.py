#!/usr/bin/env python3
"""GPU box: randomized parity sweep of the extraction path against the CPU oracle.

Random volume shapes (including rows that are not whole 16-byte vectors and volumes large enough for the fused
blur), blob-field seeds, noise levels, descriptor modes and initial scales; every record is compared bit for bit.
usage: python tools/fuzz_parity.py [cases=60] [seed=1] [max_edge=112]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (one HIP runtime in the process)
pkg = importlib.import_module("3d_sift_cuda_amd")
import _oracle


def sweep(cases=60, seed=1, max_edge=112):
    """Returns the number of cases whose records differ from the oracle's."""
    rng = np.random.default_rng(seed)
    orc = _oracle.load()
    bad = 0
    t0 = time.time()
    fields = ("x", "y", "z", "scale", "ori", "eigs", "info", "desc")
    for i in range(cases):
        dims = tuple(int(v) for v in rng.integers(12, max_edge + 1, 3))
        if i % 7 == 3:
            dims = (168, 164, 160)[i % 3:] + (168, 164, 160)[:i % 3]   # >= 2^22 voxels: the fused blur
        vseed = int(rng.integers(1, 1 << 30))
        mode = int(rng.integers(0, 4))
        init_scale = float(rng.choice([1.0, 1.0, 0.5, 2.0]))
        noise = float(rng.choice([0.0, 0.0, 1.0, 8.0]))
        fused = int(rng.choice([1, 1, 2, 0]))   # 2: the fused blur on every octave it supports, 0: never
        kp_chunks = int(rng.choice([0, 0, 0, 1, 3, 6]))   # the per-keypoint stage in chunks on two streams
        tile = int(rng.choice([0, 0, 1, 2]))        # round 4: the fused blur's tile (64 x 32 / 128 x 16)
        host_recs = int(rng.choice([5, 5, 1, 12]))  # round 4: records per candidate the pinned buffers start with (1: they grow)
        rows = int(rng.choice([0, 0, 1, 2]))        # the fused blur's thread mapping (2 on a small volume: the level-3 launch carries the subsample)
        sub = int(rng.choice([1, 1, 1, 0]))         # round 4: the subsample inside the level-3 launch / a launch of its own
        split = int(rng.choice([1, 1, 0, 2]))       # round 4: the candidate list in two parts (2: the second part overflows -> fall-back)
        dseg = int(rng.choice([32, 32, 0, 1, 5]))   # round 4: the descriptor kernel's record order over the XCDs
        order = int(rng.choice([0, 0, 1, 2, 3]))    # round 5: which workgroup takes which tile of the fused blur
        runs = int(rng.choice([0, 0, 1, 3]))        # round 5: the volume handed over in runs of planes (sift3d_set_volume_begin / _planes / _end)
        stagger = int(rng.choice([0, 0, 1, 2]))     # round 6: the fused blur's half-step stagger (0 by measurement, 1 off, 2 on)
        vol = pkg.synth_blobs(*dims, seed=vseed)
        if noise:
            vol = vol + (rng.standard_normal(vol.shape) * noise).astype(np.float32)
        with pkg.Context(*dims) as ctx:
            ctx.set_tuning(pkg.TUNE_BLUR_FUSED, fused)
            ctx.set_tuning(pkg.TUNE_KP_CHUNKS, kp_chunks)
            ctx.set_tuning(pkg.TUNE_FUSED_TILE, tile)
            ctx.set_tuning(pkg.TUNE_HOST_RECORDS, host_recs)
            ctx.set_tuning(pkg.TUNE_FUSED_ROWS, rows)
            ctx.set_tuning(pkg.TUNE_FUSED_SUB, sub)
            ctx.set_tuning(pkg.TUNE_SPLIT_TAIL, split)
            ctx.set_tuning(pkg.TUNE_DESC_SEGMENT, dseg)
            ctx.set_tuning(pkg.TUNE_FUSED_ORDER, order)
            ctx.set_tuning(pkg.TUNE_FUSED_STAGGER, stagger)
            if i % 5 == 4:
                ctx.reserve(int(rng.integers(0, 4000)))   # round 5: buffers made ahead of the run (more or fewer than it needs)
            if runs == 0:
                ctx.set_volume(vol)
            else:
                nzv = vol.shape[0]
                cuts = sorted(set([0, nzv] + [int(v) for v in rng.integers(1, nzv, runs)]))
                pieces = [(cuts[k], cuts[k + 1] - cuts[k]) for k in range(len(cuts) - 1)]
                rng.shuffle(pieces)
                ctx.set_volume_in_runs(vol, pieces)
            got = ctx.extract(initial_image_scale=init_scale, desc_mode=mode)
        want, _ = orc.extract(vol, init_scale=init_scale, desc_mode=mode)
        ok = len(got) == len(want) and all((got[f].view(np.uint32) == want[f].view(np.uint32)).all() for f in fields)
        if not ok:
            bad += 1
            where = "count" if len(got) != len(want) else ",".join(
                f for f in fields if not (got[f].view(np.uint32) == want[f].view(np.uint32)).all())
            print("MISMATCH case %d dims %s seed %d mode %d init %.1f noise %.1f fused %s rows %d sub %d tile %d split %d: %d / %d records, differs in %s"
                  % (i, dims, vseed, mode, init_scale, noise, fused, rows, sub, tile, split, len(got), len(want), where), flush=True)
        elif i % 10 == 0:
            print("case %d dims %s mode %d: %d records identical (%.0f s)" % (i, dims, mode, len(got), time.time() - t0), flush=True)
    print("%d cases, %d mismatches, %.0f s" % (cases, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:4]]
    sys.exit(1 if sweep(*a) else 0)
