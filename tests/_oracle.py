"""ctypes binding of oracle/_build/libsift3d_oracle.so (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "_build", "libsift3d_oracle.so")
LIB_OMP = os.path.join(ODIR, "_build", "libsift3d_oracle_omp.so")   # the multi-core build: bench.py's "fair CPU" line
CLI = os.path.join(ODIR, "_build", "featExtract_oracle")
# the restatement with the arithmetic of the reference's shipped CPU binary (-DO3_REFBIN_VARIANT: oracle/Makefile)
LIB_REFBIN = os.path.join(ODIR, "_build", "libsift3d_oracle_refbin.so")
CLI_REFBIN = os.path.join(ODIR, "_build", "featExtract_oracle_refbin")

# the -w / -ws test case: an anisotropic blob field with a qform (qfac -1) and an sform that differ
WORLD_CASE = dict(dims=(96, 80, 56), seed=4242, voxel=(1.0, 1.25, 1.5),
                  qform=(0.1, 0.2, 0.3, -30.0, 20.0, 5.0, -1.0),
                  sform=(0.9, 0.1, 0.0, -10.0, -0.1, 1.1, 0.05, 7.0, 0.0, -0.05, 1.4, 3.0))


def world_case_args(path):
    """argv of `featExtract_oracle --synth` that writes WORLD_CASE to path."""
    w = WORLD_CASE
    return [CLI, "--synth"] + [str(v) for v in w["dims"]] + [str(w["seed"]), path] + \
        [repr(float(v)) for v in w["voxel"] + w["qform"] + w["sform"]]
REF = os.path.join(ODIR, "_ref", "libref_partial.so")

EXT = np.dtype([("x", "<i4"), ("y", "<i4"), ("z", "<i4"), ("value", "<f4")])
REC = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("scale", "<f4"), ("ori", "<f4", (9,)), ("eigs", "<f4", (3,)),
                ("info", "<u4"), ("desc", "<f4", (64,))])
FEAT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("scale", "<f4"), ("ori", "<f4", (9,)), ("eigs", "<f4", (3,)),
                 ("info", "<u4"), ("pc", "<f4", (64,)), ("data", "<f4", (1331,))])
CAND = np.dtype([("octave", "<i4"), ("level", "<i4"), ("is_max", "<i4"), ("x", "<i4"), ("y", "<i4"), ("z", "<i4"),
                 ("value", "<f4"), ("h_value", "<f4"), ("l_value", "<f4")])


class Stats(C.Structure):
    _fields_ = [("n_octaves", C.c_int64), ("n_extrema", C.c_int64), ("n_keypoints", C.c_int64), ("t_blur", C.c_double),
                ("t_dog", C.c_double), ("t_subsample", C.c_double), ("t_detect", C.c_double), ("t_features", C.c_double),
                ("t_desc", C.c_double)]


def build():
    r = subprocess.run(["make", "-C", ODIR], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


class Oracle:
    def __init__(self, lib):
        self.L = lib
        P, I64, F, I = C.c_void_p, C.c_int64, C.c_float, C.c_int
        def sig(name, res, *a):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = list(a)
        sig("o3_gauss_filter_size", I, F, F)
        sig("o3_gauss_taps", I, F, F, P)
        sig("o3_gauss_taps_raw", I, F, F, P, I)
        sig("o3_filter3d", None, P, P, I64, I64, I64, P, I)
        sig("o3_blur", I, P, P, I64, I64, I64, F, F)
        sig("o3_dog", None, P, P, P, I64)
        sig("o3_subsample", None, P, I64, I64, I64, P)
        sig("o3_double_size", None, P, I64, I64, I64, P)
        sig("o3_halve_center", None, P, I64, I64, I64, P)
        sig("o3_detect", I, P, P, I64, I64, I64, P, I64, P, P, I64, P)
        sig("o3_detect3", I, P, P, P, I64, I64, I64, P, I64, P, P, I64, P)
        sig("o3_svd3", None, P, P, P)
        sig("o3_sort_eig", None, P, P)
        sig("o3_invert3", None, P, P)
        sig("o3_sort_high_low", None, P, I)
        sig("o3_rank", None, P)
        sig("o3_extract", I, P, I64, I64, I64, F, I, F, F, P, P, P)
        sig("o3_pyramid_features", I, P, I64, I64, I64, F, F, P, P, P)
        sig("o3_pyramid_candidates", I, P, I64, I64, I64, F, P, P)
        sig("o3_octave_levels", I, P, I64, I64, I64, P, P)
        sig("o3_free", None, P)
        sig("o3_write_key", I, C.c_char_p, P, I64, F, I, P)
        sig("o3_knn64", I, P, I64, P, I64, I, P, P)
        sig("o3_match_votes", I, P, I, P, I, P, P, I, P, P)

    @staticmethod
    def _f(a):
        return np.ascontiguousarray(a, np.float32)

    def taps(self, sigma, min_value=0.01, normalise=True):
        t = np.zeros(129, np.float32)
        n = self.L.o3_gauss_taps_raw(float(sigma), float(min_value), t.ctypes.data, 1 if normalise else 0)
        return t[:n].copy()

    def blur(self, vol, sigma, min_value=0.01):
        vol = self._f(vol); nz, ny, nx = vol.shape
        out = np.empty_like(vol)
        assert self.L.o3_blur(vol.ctypes.data, out.ctypes.data, nx, ny, nz, float(sigma), float(min_value)) == 1
        return out

    def dog(self, a, b):
        a, b = self._f(a), self._f(b)
        out = np.empty_like(a)
        self.L.o3_dog(a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
        return out

    def subsample(self, vol):
        vol = self._f(vol); nz, ny, nx = vol.shape
        out = np.empty((nz // 2, ny // 2, nx // 2), np.float32)
        self.L.o3_subsample(vol.ctypes.data, nx, ny, nz, out.ctypes.data)
        return out

    def double_size(self, vol):
        vol = self._f(vol); nz, ny, nx = vol.shape
        out = np.empty((2 * nz, 2 * ny, 2 * nx), np.float32)
        self.L.o3_double_size(vol.ctypes.data, nx, ny, nz, out.ctypes.data)
        return out

    def halve(self, vol):
        vol = self._f(vol); nz, ny, nx = vol.shape
        out = np.empty((nz // 2, ny // 2, nx // 2), np.float32)
        self.L.o3_halve_center(vol.ctypes.data, nx, ny, nz, out.ctypes.data)
        return out

    def _lists(self, fn, args, n):
        cap = n // 4 + 1024
        mins = np.zeros(cap, EXT); maxs = np.zeros(cap, EXT)
        a, b = C.c_int64(0), C.c_int64(0)
        rc = fn(*args, mins.ctypes.data, cap, C.byref(a), maxs.ctypes.data, cap, C.byref(b))
        assert rc == 0
        return mins[:a.value].copy(), maxs[:b.value].copy()

    def detect(self, H, Cc):
        H, Cc = self._f(H), self._f(Cc); nz, ny, nx = Cc.shape
        return self._lists(self.L.o3_detect, (H.ctypes.data, Cc.ctypes.data, nx, ny, nz), Cc.size)

    def detect3(self, Dp, Dc, Dn):
        Dp, Dc, Dn = self._f(Dp), self._f(Dc), self._f(Dn); nz, ny, nx = Dc.shape
        return self._lists(self.L.o3_detect3, (Dp.ctypes.data, Dc.ctypes.data, Dn.ctypes.data, nx, ny, nz), Dc.size)

    def _take(self, ptr, n, dt):
        try:
            if n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dt, n).copy()
        finally:
            self.L.o3_free(ptr)

    def extract(self, vol, init_scale=1.0, desc_mode=0, eig_thres=140.0, size_factor=1.0):
        vol = self._f(vol); nz, ny, nx = vol.shape
        out, n, st = C.c_void_p(), C.c_int64(0), Stats()
        self.L.o3_extract(vol.ctypes.data, nx, ny, nz, float(init_scale), int(desc_mode), float(eig_thres), float(size_factor),
                          C.byref(out), C.byref(n), C.byref(st))
        return self._take(out, n.value, REC), st

    def pyramid_features(self, vol, init_scale=1.0, eig_thres=140.0):
        vol = self._f(vol); nz, ny, nx = vol.shape
        out, n, st = C.c_void_p(), C.c_int64(0), Stats()
        self.L.o3_pyramid_features(vol.ctypes.data, nx, ny, nz, float(init_scale), float(eig_thres), C.byref(out), C.byref(n), C.byref(st))
        return self._take(out, n.value, FEAT), st

    def candidates(self, vol, init_scale=1.0):
        vol = self._f(vol); nz, ny, nx = vol.shape
        out, n = C.c_void_p(), C.c_int64(0)
        self.L.o3_pyramid_candidates(vol.ctypes.data, nx, ny, nz, float(init_scale), C.byref(out), C.byref(n))
        return self._take(out, n.value, CAND)

    def octave_levels(self, g0):
        g0 = self._f(g0); nz, ny, nx = g0.shape
        G = np.empty((6, nz, ny, nx), np.float32); D = np.empty((5, nz, ny, nx), np.float32)
        self.L.o3_octave_levels(g0.ctypes.data, nx, ny, nz, G.ctypes.data, D.ctypes.data)
        return G, D

    def knn64(self, db, queries, k):
        """Brute-force nearest neighbours: (idx, dist2), each (n_q, k), ascending by (distance, index)."""
        db = np.ascontiguousarray(db, np.int8); queries = np.ascontiguousarray(queries, np.int8)
        idx = np.empty((len(queries), k), np.int32); d2 = np.empty((len(queries), k), np.int32)
        assert self.L.o3_knn64(db.ctypes.data, len(db), queries.ctypes.data, len(queries), k, idx.ctypes.data, d2.ctypes.data) == 0
        return idx, d2

    def match_votes(self, first, labels, n_labels, nn_idx, nn_dist2):
        first = np.ascontiguousarray(first, np.int64); labels = np.ascontiguousarray(labels, np.int32)
        nn_idx = np.ascontiguousarray(nn_idx, np.int32); nn_dist2 = np.ascontiguousarray(nn_dist2, np.int32)
        n_img = len(first) - 1
        votes = np.zeros((n_img, n_labels), np.float32); counts = np.zeros((n_img, n_labels), np.int32)
        assert self.L.o3_match_votes(first.ctypes.data, n_img, labels.ctypes.data, n_labels, nn_idx.ctypes.data, nn_dist2.ctypes.data,
                                     nn_idx.shape[1], votes.ctypes.data, counts.ctypes.data) == 0
        return votes, counts

    def write_key(self, path, recs, eig_thres=140.0, comments=()):
        recs = np.ascontiguousarray(recs, REC)
        arr = (C.c_char_p * max(1, len(comments)))(*[c.encode() for c in comments])
        assert self.L.o3_write_key(os.fsencode(path), recs.ctypes.data, len(recs), float(eig_thres), len(comments), C.cast(arr, C.c_void_p)) == 0


_inst = None


def load():
    global _inst
    if _inst is None:
        if not os.path.exists(LIB):
            build()
        _inst = Oracle(C.CDLL(LIB))
    return _inst


_inst_omp = None


def load_omp():
    """The OpenMP build of the same restatement (bit-identical output, checked by tests/test_oracle_pins.py).  Only
    bench.py's multi-core CPU baseline uses it; the serial library stays the checker."""
    global _inst_omp
    if _inst_omp is None:
        if not os.path.exists(LIB_OMP):
            build()
        _inst_omp = Oracle(C.CDLL(LIB_OMP))
    return _inst_omp


_inst_refbin = None


def load_refbin():
    """The -DO3_REFBIN_VARIANT build: the restatement with the one arithmetic difference the shipped CPU binary of the reference
    shows in its disassembly (exp(float) is the C exp(double) in GaussianMask.cpp).  tests/test_oracle_pins.py holds it to the
    binary's own .key files byte for byte; nothing else may use it."""
    global _inst_refbin
    if _inst_refbin is None:
        if not os.path.exists(LIB_REFBIN):
            build()
        _inst_refbin = Oracle(C.CDLL(LIB_REFBIN))
    return _inst_refbin


def load_ref():
    """The partial reference build (oracle/_ref); None when it was never built."""
    if not os.path.exists(REF):
        return None
    r = C.CDLL(REF)
    P, F, I = C.c_void_p, C.c_float, C.c_int
    r.ref_gauss_filter_size.argtypes = [F, F]
    r.ref_gauss_taps_raw.argtypes = [F, I, P]
    for fn in (r.ref_svd3,):
        fn.argtypes = [P, P, P]
    for fn in (r.ref_sort_eig, r.ref_invert3):
        fn.argtypes = [P, P]
    r.ref_mult3.argtypes = [P, P, P]
    r.ref_sort_high_low.argtypes = [P, I]
    r.ref_sign.argtypes = [F]
    r.ref_write_key_text.argtypes = [P, I, C.c_char_p, F, I, P]
    r.ref_write_key_bin.argtypes = [P, I, C.c_char_p, F]
    r.ref_read_key_text.argtypes = [C.c_char_p, P, I, P]
    r.ref_dist_sqr_pcs.argtypes = [P, P, I]
    r.ref_dist_sqr_pcs.restype = F
    r.ref_world_orientation.argtypes = [P, P]
    r.ref_output_float_pgm.argtypes = [P, I, I, C.c_char_p]
    return r
