#!/usr/bin/env python3
"""GPU box: what `featMatchMultiple keys...` costs end to end, and what the .key reader is of it (round 5: the reader parses a file
in the writers' layout in parallel from memory instead of 15 million fscanf calls per 512^3-sized file).
    python tools/match_cli_profile.py [images=12] [n=256]
Writes `images` .key files (extractions of n^3 blob fields with different seeds), times sift3d_read_key on them with the parallel
reader and with the fscanf loop (same bits), then times the command line (which uses the parallel reader)."""
import importlib, os, subprocess, sys, tempfile, time
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
images = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = tempfile.mkdtemp()
names, total = [], 0
with pkg.Context(n, n, n) as ctx:
    for i in range(images):
        ctx.set_volume(pkg.synth_blobs(n, n, n, seed=100 + i))
        f = ctx.extract()
        p = os.path.join(d, "img%02d.key" % i)
        pkg.write_key(p, f, comments=["a", "b", "c"])
        names.append(p); total += len(f)
size = sum(os.path.getsize(p) for p in names)
print("%d .key files of %d^3 blob fields: %d records, %.0f MB of text" % (images, n, total, size / 1e6))
host = pkg.host_lib()
for mode, what in ((0, "parallel in-memory reader"), (1, "fscanf loop (the reference's reader restated)")):
    host.sift3d_read_key_mode(mode)
    t0 = time.perf_counter()
    blobs = []
    for p in names:
        ptr, cnt = C.c_void_p(), C.c_int64(0)
        assert host.sift3d_read_key(os.fsencode(p), C.byref(ptr), C.byref(cnt)) == 0
        blobs.append(C.string_at(ptr, cnt.value * pkg.FEATURE_DTYPE.itemsize)); host.free_ptr(ptr)
    dt = time.perf_counter() - t0
    print("sift3d_read_key over the %d files, %-48s %.3f s (%.0f MB/s)" % (images, what + ":", dt, size / dt / 1e6))
    if mode == 0:
        first = blobs
    else:
        print("same bits from both readers:", first == blobs)
host.sift3d_read_key_mode(0)
exe = os.path.join(os.path.dirname(pkg.FEATEXTRACT), "featMatchMultiple")
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-n", "5", "-o", os.path.join(d, "report")] + names, capture_output=True, text=True, cwd=d)
    print("featMatchMultiple -n 5 over the %d files, run %d: %.3f s wall (rc %d)" % (images, rep, time.perf_counter() - t0, r.returncode))
