#!/usr/bin/env python3
"""Development aid (GPU box): tiny and thin volumes through detect/extract against the oracle."""
import sys, importlib
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import _oracle
pkg = importlib.import_module("3d_sift_cuda_amd")
orc = _oracle.load()
rng = np.random.default_rng(3)
for dims in [(3,3,3),(4,4,4),(5,5,5),(12,10,9),(40,8,6),(9,33,7),(16,16,16),(31,29,27),(130,12,11)]:
    vol = pkg.synth_blobs(*dims, seed=7) + (rng.standard_normal(dims[::-1]) * 3).astype(np.float32)
    try:
        with pkg.Context(*dims) as ctx:
            ctx.set_volume(vol)
            got_c = ctx.detect(); got = ctx.extract()
        want_c = orc.candidates(vol); want,_ = orc.extract(vol)
        ok = len(got)==len(want) and len(got_c)==len(want_c) and (len(got)==0 or ((got["desc"]==want["desc"]).all() and (got["x"].view(np.uint32)==want["x"].view(np.uint32)).all()))
        print(dims, "cands", len(got_c), len(want_c), "recs", len(got), len(want), "OK" if ok else "MISMATCH")
    except Exception as e:
        print(dims, "EXC", repr(e)[:200])
