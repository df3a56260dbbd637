#!/usr/bin/env python3
"""GPU box: BASELINE configurations through the drop-in command line itself, with its per-phase times (SIFT3D_CLI_TIMES).
    python tools/cli_configs.py
  C2  256^3 .nii, SIFT-rank                      featExtract -d0 v256.nii out.key
  C3  512^3 .nii, -2+ (1024^3 processed), BRIEF  featExtract -d0 -2+ -b v512.nii out.key
  and the metric volume cut into two Z-slabs on one device (rehearsal of -d0,1): featExtract -d0,0 v512.nii out.key"""
import importlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
d = tempfile.mkdtemp()
files = {}
for n in (256, 512):
    files[n] = os.path.join(d, "v%d.nii" % n)
    pkg.write_nifti(files[n], pkg.synth_blobs(n, n, n, seed=12345))
runs = [("C2: featExtract -d0 v256.nii", ["-d0", files[256]]),
        ("C3: featExtract -d0 -2+ -b v512.nii", ["-d0", "-2+", "-b", files[512]]),
        ("metric volume: featExtract -d0 v512.nii", ["-d0", files[512]]),
        ("two slabs on one device: featExtract -d0,0 v512.nii", ["-d0,0", files[512]])]
for name, args in runs:
    for rep in range(2):
        key = os.path.join(d, "out.key")
        t0 = time.perf_counter()
        r = subprocess.run([pkg.FEATEXTRACT] + args + [key], capture_output=True, text=True, env=dict(os.environ, SIFT3D_CLI_TIMES="1"))
        wall = time.perf_counter() - t0
        recs = int(open(key).read(400).split("Features: ")[1].split("\n")[0]) if r.returncode == 0 else -1
        print("%-52s run %d: %.3f s wall, rc %d, %d records, .key %.0f MB" % (name, rep, wall, r.returncode, recs, os.path.getsize(key) / 1e6 if recs >= 0 else 0))
        if rep == 1:
            print("    " + " | ".join(l[2:] for l in r.stderr.splitlines() if l.startswith("# ")))
