#!/usr/bin/env python3
"""GPU box: ms per extraction of a resident 512^3 volume by the number of chunks of the per-keypoint stage
(SIFT3D_TUNE_KP_CHUNKS: keypoint kernel of chunk i+1 beside the descriptor kernel of chunk i) and by descriptor mode.
usage: python tools/kp_chunks.py [N=512] [reps=10]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = pkg.Context(n, n, n)
d = torch.from_numpy(pkg.synth_blobs(n, n, n)).cuda()
torch.cuda.synchronize()
ctx.set_volume_dev(d.data_ptr(), n, n, n)
ctx.sync()
want = None
for mode in (0, 2):
    for chunks in (1, 2, 3, 4, 6, 8, 12, 16, 0):
        ctx.set_tuning(pkg.TUNE_KP_CHUNKS, chunks)
        for _ in range(3):
            f = ctx.extract(desc_mode=mode, copy=False)
        if want is None or chunks == 1:
            want = f.tobytes()
        same = f.tobytes() == want
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.extract(desc_mode=mode, copy=False)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("mode %d chunks %2d: %.3f ms per extraction, %d records, same bytes as one chunk: %s" % (mode, chunks, ms, len(f), same), flush=True)
