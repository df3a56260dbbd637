#!/usr/bin/env python3
"""Where the time of reading a .nii.gz goes on this box: file read, inflate (libdeflate in one call / zlib), page faults of the
buffers, the copy into the caller's array.  usage: python tools/inflate_probe.py [N=512]"""
import ctypes, importlib, os, subprocess, sys, tempfile, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = tempfile.mkdtemp()
vol = pkg.synth_blobs(n, n, n, seed=12345)
nii, gz = os.path.join(d, "v.nii"), os.path.join(d, "v.nii.gz")
pkg.write_nifti(nii, vol)
subprocess.run("gzip -1 -c %s > %s" % (nii, gz), shell=True, check=True)
t = time.perf_counter(); comp = open(gz, "rb").read(); print("read %d MB compressed: %.3f s" % (len(comp) >> 20, time.perf_counter() - t))
raw_n = os.path.getsize(nii)
L = ctypes.CDLL("libdeflate.so.0")
L.libdeflate_alloc_decompressor.restype = ctypes.c_void_p
L.libdeflate_gzip_decompress.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
dec = L.libdeflate_alloc_decompressor()
for touched in (False, True, True):
    buf = np.empty(raw_n, np.uint8)
    if touched:
        t = time.perf_counter(); buf[::4096] = 0; print("  first touch of %d MB: %.3f s" % (raw_n >> 20, time.perf_counter() - t))
    got = ctypes.c_size_t(0)
    t = time.perf_counter(); rc = L.libdeflate_gzip_decompress(dec, comp, len(comp), buf.ctypes.data_as(ctypes.c_void_p), raw_n, ctypes.byref(got)); dt = time.perf_counter() - t
    print("libdeflate into a%s buffer: %.3f s (%.0f MB/s) rc %d" % (" touched" if touched else "n untouched", dt, raw_n / 1e6 / dt, rc))
t = time.perf_counter(); out = zlib.decompress(comp, 31); dt = time.perf_counter() - t
print("zlib.decompress: %.3f s (%.0f MB/s)" % (dt, raw_n / 1e6 / dt))
dst = np.empty(raw_n, np.uint8); dst[::4096] = 0
t = time.perf_counter(); dst[:] = buf; print("copy %d MB: %.3f s" % (raw_n >> 20, time.perf_counter() - t))
for on in (1, 0):
    pkg.nifti_fast_inflate(on)
    t = time.perf_counter(); v, _ = pkg.read_nifti(gz); print("read_nifti fast=%d: %.3f s" % (on, time.perf_counter() - t))
