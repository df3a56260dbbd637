#!/usr/bin/env python3
"""Development aid: time phase A of the per-keypoint stage cut after stage N (SIFT3D_KP_STOP)."""
import importlib, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    pkg = importlib.import_module("3d_sift_cuda_amd")
    n = int(os.environ.get("ABL_N", "256"))
    ctx = pkg.Context(n, n, n); ctx.set_volume(pkg.synth_blobs(n, n, n))
    ctx.extract(); ctx.enable_timing(True); ctx.extract()
    log = ctx.launch_log(); sel = log[log["stage"] == 5]
    print("stop=%s keypoint ms: first3 %s  total %.3f  (items %s)" % (os.environ.get("SIFT3D_KP_STOP", "0"), [round(float(x), 3) for x in sel["ms"][:3]], sel["ms"].sum(), sel["nvox"][:3].tolist()))
else:
    for stop in [int(v) for v in os.environ.get("ABL_STOPS", "1,2,3,4,7,8,31,32,33,34,0").split(",")]:
        env = dict(os.environ, SIFT3D_KP_STOP=str(stop))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
