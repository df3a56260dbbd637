#!/usr/bin/env python3
"""GPU box: what `featExtract in.nii out.key` costs end to end (round-4 review item 2).

    python tools/cli_profile.py [sizes=256,512] [reps=3] [oracle_sizes=256,512]

For every size: a blob-field volume written as .nii and .nii.gz (float32), then `SIFT3D_CLI_TIMES=1 featExtract -d0`
on each, `reps` times, the per-phase wall times the CLI prints and the process's whole wall time (`hip init` and `context`
run in a thread of their own BESIDE `read`; `waited` is what the main thread still waited for them after the read; `upload`
is what was left to upload then; `main` is main()'s own duration: wall - main = process start, library loading, exit); beside them
the oracle's CLI (`oracle/_build/featExtract_oracle`, the CPU restatement behind the same command line: TEST infrastructure,
timed here as the CPU side of the same box).  The .key files of the two are compared byte for byte.
"""
import hashlib
import numpy as np
import importlib
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_sift_cuda_amd")
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,512").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
osizes = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "256,512").split(",") if v]
ORACLE = os.path.join(ROOT, "oracle", "_build", "featExtract_oracle")
PHASES = ["read image", "hip runtime", "device context", "waited", "upload", "extraction", "write features", "teardown", "main"]


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 22), b""):
            h.update(b)
    return h.hexdigest()[:16]


def run_cli(args, env=None):
    t0 = time.perf_counter()
    r = subprocess.run(args, capture_output=True, text=True, env=env)
    wall = time.perf_counter() - t0
    ph = {}
    for m in re.finditer(r"^# ([a-z ]+): ([0-9.]+) s", r.stderr, re.M):
        ph[m.group(1)] = float(m.group(2))
    m = re.search(r"device context: ([0-9.]+) s .*waited ([0-9.]+) s", r.stderr)   # "# hip runtime: a s, device context: b s (...); waited c s ..."
    if m:
        ph["device context"], ph["waited"] = float(m.group(1)), float(m.group(2))
    return r.returncode, wall, ph, r


d = tempfile.mkdtemp()
print("featExtract end to end on this box (%d host cores visible); times in seconds" % (os.cpu_count() or 0))
for n in sizes:
    vol = pkg.synth_blobs(n, n, n, seed=12345)
    nii, gz = os.path.join(d, "v%d.nii" % n), os.path.join(d, "v%d.nii.gz" % n)
    pkg.write_nifti(nii, vol)
    subprocess.run("gzip -1 -c %s > %s" % (nii, gz), shell=True, check=True)
    print("\n== %d^3 float32: .nii %.0f MB, .nii.gz %.0f MB ==" % (n, os.path.getsize(nii) / 1e6, os.path.getsize(gz) / 1e6))
    print("%-28s %8s | %s" % ("run", "wall", " ".join("%9s" % p.replace("device context", "context").replace("write features", "write").replace("hip runtime", "hip init").replace("read image", "read") for p in PHASES)))
    keys = {}
    for src, tag in ((nii, ".nii"), (gz, ".nii.gz")):
        key = os.path.join(d, "out%d%s.key" % (n, tag.replace(".", "_")))
        for rep in range(reps):
            rc, wall, ph, r = run_cli([pkg.FEATEXTRACT, "-d0", src, key], env=dict(os.environ, SIFT3D_CLI_TIMES="1"))
            if rc != 0:
                print("featExtract failed:", r.stdout[-300:], r.stderr[-300:])
                sys.exit(1)
            print("%-28s %8.3f | %s" % ("featExtract -d0 %s #%d" % (tag, rep), wall, " ".join("%9.3f" % ph.get(p, float("nan")) for p in PHASES)))
        keys[tag] = sha(key)
        print("%-28s .key %.1f MB sha %s" % ("", os.path.getsize(key) / 1e6, keys[tag]))
    if n in osizes and os.path.exists(ORACLE):
        key = os.path.join(d, "oracle%d.key" % n)
        rc, wall, ph, r = run_cli([ORACLE, nii, key])
        print("%-28s %8.3f | (CPU restatement, one thread; rc %d) .key sha %s%s" % ("featExtract_oracle .nii", wall, rc, sha(key) if rc == 0 else "-",
                                                                                  "  == GPU CLI bytes" if rc == 0 and sha(key) == keys[".nii"] else "  DIFFERENT"))
    assert keys[".nii"] == keys[".nii.gz"]
    # the reader alone on the gzip'ed file: one call of libdeflate (default where the system has the library) against zlib's stream
    i16 = os.path.join(d, "v%d_i16.nii.gz" % n)
    h = bytearray(open(nii, "rb").read(352))
    h[70:72] = (4).to_bytes(2, "little"); h[72:74] = (16).to_bytes(2, "little")     # DT_INT16: what scanners write
    import gzip
    with gzip.open(i16, "wb", compresslevel=1) as f:
        f.write(bytes(h)); f.write(np.clip(vol * 8.0 + 100.0, -32000, 32000).astype(np.int16).tobytes())
    for path, tag in ((gz, "float32 .nii.gz"), (i16, "int16 .nii.gz (%.0f MB)" % (os.path.getsize(i16) / 1e6))):
        for on in (1, 0, 1, 0):
            pkg.nifti_fast_inflate(on)
            c0 = pkg.nifti_fast_inflate_count()
            t0 = time.perf_counter(); v, _ = pkg.read_nifti(path); dt = time.perf_counter() - t0
            print("%-28s %8.3f | read_nifti alone, %s" % (tag, dt, "libdeflate, one call" if pkg.nifti_fast_inflate_count() > c0 else "zlib, streaming"))
    pkg.nifti_fast_inflate(1)
