#!/usr/bin/env python3
"""GPU box: does ending the command line with _exit (default) instead of an orderly teardown (SIFT3D_CLI_CLEAN_EXIT=1) push its cost onto the
NEXT process?  The same command several times back to back, both ways, wall time of each run and its context / extraction phases.
usage: python tools/cli_exit_probe.py"""
import importlib, os, re, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
d = tempfile.mkdtemp()
nii = os.path.join(d, "v512.nii")
pkg.write_nifti(nii, pkg.synth_blobs(512, 512, 512, seed=12345))
key = os.path.join(d, "out.key")
for label, args in (("metric volume (7 GB context)", ["-d0", nii]), ("C3: -2+ -b (58 GB context)", ["-d0", "-2+", "-b", nii])):
    for clean in (0, 1, 0, 1):
        env = dict(os.environ, SIFT3D_CLI_TIMES="1")
        if clean:
            env["SIFT3D_CLI_CLEAN_EXIT"] = "1"
        row = []
        for rep in range(4):
            t0 = time.perf_counter()
            r = subprocess.run([pkg.FEATEXTRACT] + args + [key], capture_output=True, text=True, env=env)
            wall = time.perf_counter() - t0
            m = re.search(r"device context: ([0-9.]+) s", r.stderr); c = float(m.group(1)) if m else -1
            m = re.search(r"# extraction: ([0-9.]+) s", r.stderr); e = float(m.group(1)) if m else -1
            m = re.search(r"# teardown: ([0-9.]+) s", r.stderr); t = float(m.group(1)) if m else -1
            row.append("%.2f (ctx %.2f, extr %.2f, teardown %.2f)" % (wall, c, e, t))
        print("%-30s %-14s %s" % (label, "orderly exit:" if clean else "_exit:", "  ".join(row)), flush=True)
