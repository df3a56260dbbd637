#!/usr/bin/env python3
"""Timeline of one extraction: every launch bracketed by HIP events on the stream it runs on (timing mode 1), listed with
the time it began since the first launch -- what runs beside what, and where the chip waits.
usage: python tools/timeline.py [N=512]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = pkg.synth_blobs(n, n, n)
with pkg.Context(n, n, n) as ctx:
    ctx.set_volume(vol)
    ctx.extract(); ctx.extract()
    ctx.enable_timing(1)
    f = ctx.extract()
    log = ctx.launch_log()
print("%d records; start_ms end_ms ms stage taps voxels" % len(f))
for r in sorted(log, key=lambda r: r["start_ms"]):
    print("%8.3f %8.3f %7.3f  %-12s %2d %10d" % (r["start_ms"], r["start_ms"] + r["ms"], r["ms"], pkg.STAGES[r["stage"]], r["ntaps"], r["nvox"]))
