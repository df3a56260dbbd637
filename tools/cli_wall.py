#!/usr/bin/env python3
"""GPU box: wall time of the featExtract command line on a 512^3 .nii (read + upload + extraction + .key text)."""
import importlib, os, subprocess, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = tempfile.mkdtemp()
nii, key = os.path.join(d, "v.nii"), os.path.join(d, "v.key")
t0 = time.time(); vol = pkg.synth_blobs(n, n, n, seed=12345); pkg.write_nifti(nii, vol); print("synth + write .nii %.2f s" % (time.time() - t0))
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([pkg.FEATEXTRACT, "-d0", nii, key], capture_output=True, text=True)
    print("featExtract run %d: %.2f s (rc %d), .key %.1f MB" % (rep, time.time() - t0, r.returncode, os.path.getsize(key) / 1e6))
env = dict(os.environ, SIFT3D_CLI_TIMES="1")
r = subprocess.run([pkg.FEATEXTRACT, "-d0", nii, key], capture_output=True, text=True, env=env)
print(r.stderr[-600:])
