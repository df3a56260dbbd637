#!/usr/bin/env python3
"""Development aid: time the descriptor kernel cut after stage N (SIFT3D_KP_STOP = 11..15)."""
import importlib, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    pkg = importlib.import_module("3d_sift_cuda_amd")
    n = int(os.environ.get("ABL_N", "256"))
    ctx = pkg.Context(n, n, n); ctx.set_volume(pkg.synth_blobs(n, n, n))
    ctx.extract(); ctx.enable_timing(True); ctx.extract()
    log = ctx.launch_log(); sel = log[log["stage"] == 6]
    print("stop=%s descriptor ms %.3f (records %s)" % (os.environ.get("SIFT3D_KP_STOP", "0"), sel["ms"].sum(), sel["nvox"].tolist()))
else:
    for stop in [int(v) for v in os.environ.get("ABL_STOPS", "21,11,12,13,14,0").split(",")]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SIFT3D_KP_STOP=str(stop)))
