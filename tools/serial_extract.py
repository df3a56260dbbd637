#!/usr/bin/env python3
"""A few extractions with every launch on the context's main stream (timing mode 3: nothing shares the chip with a
kernel), for a rocprofv3 kernel trace / PMC pass whose per-kernel numbers are that kernel's own.
usage: python tools/serial_extract.py [N=512] [reps=3]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
pkg = importlib.import_module("3d_sift_cuda_amd")
if os.environ.get("SIFT3D_BUILD"):   # an A/B build of the library beside the product's: csrc/<SIFT3D_BUILD>/libsift3d_hip.so
    pkg.LIB_HIP = os.path.join(pkg.CSRC, os.environ["SIFT3D_BUILD"], "libsift3d_hip.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vol = pkg.synth_blobs(n, n, n)
with pkg.Context(n, n, n) as ctx:
    ctx.set_volume(vol)
    ctx.extract()
    ctx.enable_timing(3)
    for _ in range(reps):
        f = ctx.extract()
    t = ctx.timings()
    print("records %d extrema %d" % (len(f), t["n_extrema"]))
    for s, v in t["stages"].items():
        if v["launches"]:
            print("%-12s %3d launches %8.3f ms" % (s, v["launches"], v["ms"]))
