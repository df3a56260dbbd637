import importlib, sys
sys.path.insert(0, "/root/repo")
import os
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("3d_sift_cuda_amd")
for n, mode in ((96, 0), (200, 0), (256, 0)):
    vol = pkg.synth_blobs(n, n, n, seed=7 + n)
    with pkg.Context(n, n, n) as ctx:
        ctx.set_volume(vol)
        a = ctx.extract(desc_mode=mode)
        ctx.set_tuning(pkg.TUNE_DESC_THREADS, 64)
        b = ctx.extract(desc_mode=mode)
        ctx.set_tuning(pkg.TUNE_SAMPLER_CAP, 8)
        c = ctx.extract(desc_mode=mode)
    print(n, len(a), "same bytes 64 vs 128:", a.tobytes() == b.tobytes(), a.tobytes() == c.tobytes())
