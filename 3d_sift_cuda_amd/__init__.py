"""3d_sift_cuda_amd -- MI355X-native 3D SIFT extraction path (Python mirror of the C-ABI).

Thin ctypes layer over ``csrc/_build/libsift3d_hip.so`` (``include/sift3d.h``).  The
names follow the reference's operator interface for this path
(R/cuda_common/SIFT_cuda_Tools.cuh, R/src_common/MultiScale.h; R/ =
/root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/):

=====================  =====================================================================
here                   reference entry point it replaces
=====================  =====================================================================
``gauss_blur``         gb3d_blur3d -> blur_3d_simpleborders_CUDA_Row_Col_Shared_mem (.cuh:69-76)
``dog``                fioMultSum_interleave(.., -1.0f) -> fioCudaMultSum (.cuh:213-217)
``subsample2``         Subsample_interleave -> SubSampleInterpolateCuda (.cuh:202-205)
``extrema``            detectExtrema4D_test_interleave -> detectExtrema4D_test_cuda (.cuh:32-38)
``detect``/``extract`` msGeneratePyramidDOG3D_efficient (MultiScale.h:534-543) + main()'s descriptor loop
=====================  =====================================================================

There is no CPU fallback: importing works everywhere (the library is only
dlopen'ed on first use), but every compute call raises if the HIP library is
missing or no HIP device is usable.  The package name starts with a digit, so
load it with ``importlib.import_module("3d_sift_cuda_amd")``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_HIP = os.path.join(CSRC, "_build", "libsift3d_hip.so")
LIB_HOST = os.path.join(CSRC, "_build", "libsift3d_host.so")
FEATEXTRACT = os.path.join(CSRC, "_build", "featExtract")

DESC_SIFT, DESC_BRIEF, DESC_RRIEF, DESC_NRRIEF = 0, 1, 2, 3
ABI_VERSION = 6   # SIFT3D_ABI_VERSION of include/sift3d.h: the structure layouts this file mirrors
INFO_MIN0MAX1, INFO_REORIENT = 0x10, 0x20
STAGES = ("blur_x", "blur_y", "blur_z_dog", "subsample", "extrema", "keypoint", "descriptor", "blur_fused", "octave_tiny")

# sift3d_tuning (include/sift3d.h)
TUNE_BLUR_FUSED, TUNE_FUSED_CHUNKS, TUNE_FUSED_ROWS, TUNE_LAZY_LEVELS, TUNE_TINY_OCTAVE, TUNE_SAMPLER_CAP, TUNE_KP_CHUNKS, TUNE_BANDS_FIRST, TUNE_HOST_RECORDS, TUNE_FUSED_TILE, TUNE_FUSED_SUB, TUNE_SPLIT_TAIL, TUNE_DESC_SEGMENT, TUNE_FUSED_ORDER, TUNE_FUSED_STAGGER = range(15)

EXTREMUM_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("z", "<i4"), ("value", "<f4")])
FEATURE_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("scale", "<f4"), ("ori", "<f4", (9,)),
                          ("eigs", "<f4", (3,)), ("info", "<u4"), ("desc", "<f4", (64,))])
CANDIDATE_DTYPE = np.dtype([("octave", "<i4"), ("level", "<i4"), ("is_max", "<i4"), ("x", "<i4"), ("y", "<i4"),
                            ("z", "<i4"), ("value", "<f4"), ("h_value", "<f4"), ("l_value", "<f4")])


class Sift3DError(RuntimeError):
    pass


class _Timings(C.Structure):
    _fields_ = [("ms", C.c_double * len(STAGES)), ("launches", C.c_int64 * len(STAGES)), ("alg_bytes", C.c_double * len(STAGES)),
                ("n_octaves", C.c_int64), ("n_extrema", C.c_int64), ("n_keypoints", C.c_int64),
                ("n_records", C.c_int64), ("total_ms", C.c_double)]


class LevelDesc(C.Structure):
    """sift3d_level_desc (include/sift3d.h)"""
    _fields_ = [("img", C.c_void_p), ("dogc", C.c_void_p), ("nx", C.c_int64), ("ny", C.c_int64), ("nz_local", C.c_int64),
                ("nz_global", C.c_int64), ("z_offset", C.c_int64), ("sigma_h", C.c_float), ("sigma_c", C.c_float),
                ("sigma_l", C.c_float), ("octave_factor", C.c_float)]


LAUNCH_DTYPE = np.dtype([("stage", "<i4"), ("ntaps", "<i4"), ("nvox", "<i8"), ("alg_bytes", "<f8"), ("ms", "<f8"),
                         ("start_ms", "<f8")])


def build(verbose=False):
    """Compile the HIP library, the host helpers and the CLI for gfx950 (in-tree)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise Sift3DError("building 3d_sift_cuda_amd/csrc failed")


_hip = None
_host = None


def _sig(fn, res, *args):
    fn.restype = res
    fn.argtypes = list(args)


def hip_lib():
    """dlopen libsift3d_hip.so and declare the C-ABI; raises if it is not built."""
    global _hip
    if _hip is not None:
        return _hip
    if not os.path.exists(LIB_HIP):
        raise Sift3DError("%s is missing: run __graft_entry__.build() (there is no CPU fallback)" % LIB_HIP)
    L = C.CDLL(LIB_HIP)
    P, I64, F, I = C.c_void_p, C.c_int64, C.c_float, C.c_int
    _sig(L.sift3d_abi_version, I)
    if L.sift3d_abi_version() != ABI_VERSION:   # the library writes whole structures through our pointers
        raise Sift3DError("%s has ABI version %d, this mirror of include/sift3d.h is version %d: rebuild (__graft_entry__.build())"
                          % (LIB_HIP, L.sift3d_abi_version(), ABI_VERSION))
    _sig(L.sift3d_device_count, I)
    _sig(L.sift3d_create, P, I, I64, I64, I64)
    _sig(L.sift3d_create_slab, P, I, I64, I64, I64)
    _sig(L.sift3d_set_tuning, I, P, I, I)
    _sig(L.sift3d_host_buffer_grows, I64, P)
    _sig(L.sift3d_zslab_set_tuning, I, P, I, I)
    _sig(L.sift3d_destroy, None, P)
    _sig(L.sift3d_last_error, C.c_char_p, P)
    _sig(L.sift3d_set_stream, I, P, P)
    _sig(L.sift3d_sync, I, P)
    _sig(L.sift3d_free, None, P)
    _sig(L.sift3d_gauss_taps, I, F, F, P)
    _sig(L.sift3d_set_libm_variant, I, I)
    _sig(L.sift3d_get_libm_variant, I)
    _sig(L.sift3d_gauss_blur, I, P, P, P, I64, I64, I64, F, F)
    _sig(L.sift3d_gauss_blur_dev, I, P, P, P, I64, I64, I64, F, F)
    _sig(L.sift3d_gauss_blur_dog_dev, I, P, P, P, P, I64, I64, I64, F, F)
    _sig(L.sift3d_gauss_blur_dog_half_dev, I, P, P, P, P, P, I64, I64, I64, F, F, C.POINTER(C.c_int))
    _sig(L.sift3d_blur_window_supported, I, I64, I64, F, F)
    _sig(L.sift3d_gauss_blur_dog_window_dev, I, P, P, P, P, I64, I64, I64, I64, I64, F, F)
    _sig(L.sift3d_dog, I, P, P, P, P, I64)
    _sig(L.sift3d_dog_dev, I, P, P, P, P, I64)
    _sig(L.sift3d_subsample2, I, P, P, I64, I64, I64, P)
    _sig(L.sift3d_subsample2_dev, I, P, P, I64, I64, I64, P)
    _sig(L.sift3d_extrema, I, P, P, P, P, I64, I64, I64, P, I64, P, P, I64, P)
    _sig(L.sift3d_double_size, I, P, P, I64, I64, I64, P)
    _sig(L.sift3d_halve_size, I, P, P, I64, I64, I64, P)
    _sig(L.sift3d_selftest_lds_add, I, P, P, P, I64, P, P)
    _sig(L.sift3d_set_volume, I, P, P, I64, I64, I64)
    _sig(L.sift3d_set_volume_dev, I, P, P, I64, I64, I64)
    _sig(L.sift3d_set_volume_resized, I, P, P, I64, I64, I64, I)
    _sig(L.sift3d_reserve, I, P, I64)
    _sig(L.sift3d_set_volume_begin, I, P, I64, I64, I64, I)
    _sig(L.sift3d_set_volume_planes, I, P, P, I64, I64)
    _sig(L.sift3d_set_volume_end, I, P)
    _sig(L.sift3d_detect, I, P, F, P, P)
    _sig(L.sift3d_extract, I, P, F, I, F, F, P, P)
    _sig(L.sift3d_extract_view, I, P, F, I, F, F, P, P)
    _sig(L.sift3d_enable_timing, I, P, I)
    _sig(L.sift3d_get_timings, I, P, P)
    _sig(L.sift3d_get_launch_log, I, P, P, I64, P)
    _sig(L.sift3d_candidates_reset, I, P)
    _sig(L.sift3d_extrema_append_dev, I, P, P, P, P, I64, I64, I64, I, I64, I64)
    _sig(L.sift3d_lazy_levels_supported, I, I64, I64, I64, F)
    _sig(L.sift3d_extrema_append_lazy_dev, I, P, P, P, P, P, P, P, F, I64, I64, I64, I, I64, I64)
    _sig(L.sift3d_candidates_dev, I, P, P, I, P, P)
    _sig(L.sift3d_describe_dev, I, P, P, I, I, F, F, P, P, P)
    _sig(L.sift3d_describe_dev_counts, I, P, P, I, I, F, F, P, P)
    _sig(L.sift3d_describe_dev_place, I, P, P, P, P, P, P)
    _sig(L.sift3d_host_register, I, P, I64)
    _sig(L.sift3d_host_unregister, I, P)
    _sig(L.sift3d_set_max_octaves, I, P, I)
    _sig(L.sift3d_extract_zslab, I, P, I, P, I64, I64, I64, F, I, F, F, P, P, P, C.c_char_p, I64)
    _sig(L.sift3d_extract_zslab_over, I, I, P, I, P, I64, I64, I64, F, I, F, F, P, P, P, C.c_char_p, I64)
    _sig(L.sift3d_zslab_set_transport_library, None, C.c_char_p)
    _sig(L.sift3d_zslab_create, P, P, I, I64, I64, I64, C.c_char_p, I64)
    _sig(L.sift3d_zslab_extract, I, P, P, F, I, F, F, P, P, P, C.c_char_p, I64)
    _sig(L.sift3d_zslab_destroy, None, P)
    _sig(L.sift3d_zslab_set_volume, I, P, P, C.c_char_p, I64)
    _sig(L.sift3d_zslab_extract_resident, I, P, F, I, F, F, P, P, P, C.c_char_p, I64)
    _sig(L.sift3d_knn64, I, I, P, I64, P, I64, I, P, P, I, P, C.c_char_p, I64)
    _sig(L.sift3d_get_level_slice, I, P, I, I, I64, P, P, P)
    _hip = L
    return L


def host_lib():
    """dlopen libsift3d_host.so (NIfTI I/O, .key writer, synthetic volumes; no GPU needed)."""
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(LIB_HOST):
        raise Sift3DError("%s is missing: run __graft_entry__.build()" % LIB_HOST)
    L = C.CDLL(LIB_HOST)
    P, I64, F, I = C.c_void_p, C.c_int64, C.c_float, C.c_int
    _sig(L.sift3d_synth_blobs, None, P, I64, I64, I64, C.c_uint32)
    _sig(L.sift3d_synth_blobs_slices, None, P, I64, I64, I64, C.c_uint32, I64, I64)
    _sig(L.nifti_min_read, I, C.c_char_p, P)
    _sig(L.nifti_min_free, None, P)
    _sig(L.nifti_min_write_f32, I, C.c_char_p, P, I, I, I, F, F, F)
    _sig(L.nifti_min_write_f32_ex, I, C.c_char_p, P, I, I, I, F, F, F, P, P)
    _sig(L.sift3d_write_key, I, C.c_char_p, P, I64, F, I, P)
    _sig(L.sift3d_write_key_mode, None, I)
    _sig(L.sift3d_write_key_bin, I, C.c_char_p, P, I64, F)
    _sig(L.sift3d_read_key, I, C.c_char_p, P, P)
    _sig(L.sift3d_read_key_mode, None, I)
    _sig(L.sift3d_write_pgm, I, C.c_char_p, P, I, I)
    _sig(L.sift3d_world_transform, None, P, I64, P)
    _sig(L.sift3d_match_filter, I64, P, I64, I, I)
    _sig(L.sift3d_match_descriptors, I, P, I64, P)
    _sig(L.sift3d_match_votes, I, P, P, I, P, I, P, P, I, P, P)
    _sig(L.sift3d_match_write_votes, I, C.c_char_p, C.c_char_p, C.c_char_p, P, P, I, I, I)
    L.free_ptr = C.CDLL(None).free
    L.free_ptr.argtypes = [C.c_void_p]
    _host = L
    return L


class ZSlabStats(C.Structure):
    """sift3d_zslab_stats"""
    _fields_ = [("n_ranks", C.c_int32), ("sharded_octaves", C.c_int32), ("exchanges", C.c_int64), ("halo_bytes_critical", C.c_int64),
                ("halo_bytes_deferred", C.c_int64), ("gather_bytes", C.c_int64), ("n_extrema", C.c_int64), ("n_keypoints", C.c_int64),
                ("n_records", C.c_int64), ("wall_ms", C.c_double), ("halo_bytes_hidden", C.c_int64), ("transport", C.c_int32),
                ("transport_fell_back", C.c_int32), ("rccl_version", C.c_int32), ("comm_sets", C.c_int32), ("resident_volume", C.c_int32),
                ("list_grown", C.c_int32), ("merge_ms", C.c_double), ("halo_bytes_subsample", C.c_int64), ("enqueue_ms", C.c_double)]


ZSLAB_TRANSPORT, TRANSPORT_PEER_COPY, TRANSPORT_RCCL = 1000, 0, 1   # sift3d_zslab_set_tuning(h, SIFT3D_ZSLAB_TRANSPORT, ...)
ZSLAB_SERIAL_CHANNELS, ZSLAB_DUPLICATE_RANKS, ZSLAB_POISON_HALO, ZSLAB_PATCH_WAIT, ZSLAB_LIST_ROOM = 1001, 1002, 1003, 1004, 1005   # include/sift3d.h


def zslab_set_transport_library(path):
    """sift3d_zslab_set_transport_library: the RCCL build the slab driver loads (None: librccl.so.1)."""
    hip_lib().sift3d_zslab_set_transport_library(None if path is None else os.fsencode(path))


def extract_zslab(vol, devices, initial_image_scale=1.0, desc_mode=DESC_SIFT, eig_thres=140.0, size_factor=1.0, transport=TRANSPORT_PEER_COPY):
    """sift3d_extract_zslab_over: the volume cut into one Z-slab per entry of `devices`, one process, halos by peer copies
    or RCCL.  Returns (records, stats dict)."""
    vol = _f32(vol)
    nz, ny, nx = vol.shape
    dev = (C.c_int * len(devices))(*[int(d) for d in devices])
    out, n, st, err = C.c_void_p(), C.c_int64(0), ZSlabStats(), C.create_string_buffer(512)
    rc = hip_lib().sift3d_extract_zslab_over(int(transport), dev, len(devices), vol.ctypes.data, nx, ny, nz, float(initial_image_scale),
                                             int(desc_mode), float(eig_thres), float(size_factor), C.byref(out), C.byref(n), C.byref(st), err, 512)
    if rc != 0:
        e = Sift3DError("sift3d_extract_zslab -> %d: %s" % (rc, err.value.decode(errors="replace")))
        e.code = rc
        raise e
    try:
        recs = np.frombuffer((C.c_char * (n.value * FEATURE_DTYPE.itemsize)).from_address(out.value), FEATURE_DTYPE, n.value).copy() if n.value else np.zeros(0, FEATURE_DTYPE)
    finally:
        hip_lib().sift3d_free(out)
    return recs, {k: getattr(st, k) for k, _ in ZSlabStats._fields_}


class ZSlab:
    """sift3d_zslab_create / _extract / _destroy: the slabs' contexts kept between volumes of one shape."""

    def __init__(self, nx, ny, nz, devices):
        self._L = hip_lib()
        dev = (C.c_int * len(devices))(*[int(d) for d in devices])
        err = C.create_string_buffer(512)
        self._h = self._L.sift3d_zslab_create(dev, len(devices), nx, ny, nz, err, 512)
        if not self._h:
            raise Sift3DError("sift3d_zslab_create: %s" % err.value.decode(errors="replace"))
        self.shape = (nz, ny, nx)

    def extract(self, vol, initial_image_scale=1.0, desc_mode=DESC_SIFT, eig_thres=140.0, size_factor=1.0):
        vol = _f32(vol)
        assert vol.shape == self.shape, (vol.shape, self.shape)
        out, n, st, err = C.c_void_p(), C.c_int64(0), ZSlabStats(), C.create_string_buffer(512)
        rc = self._L.sift3d_zslab_extract(self._h, vol.ctypes.data, float(initial_image_scale), int(desc_mode), float(eig_thres),
                                          float(size_factor), C.byref(out), C.byref(n), C.byref(st), err, 512)
        if rc != 0:
            e = Sift3DError("sift3d_zslab_extract -> %d: %s" % (rc, err.value.decode(errors="replace")))
            e.code = rc
            raise e
        try:
            recs = np.frombuffer((C.c_char * (n.value * FEATURE_DTYPE.itemsize)).from_address(out.value), FEATURE_DTYPE, n.value).copy() if n.value else np.zeros(0, FEATURE_DTYPE)
        finally:
            self._L.sift3d_free(out)
        return recs, {k: getattr(st, k) for k, _ in ZSlabStats._fields_}

    def set_volume(self, vol):
        """sift3d_zslab_set_volume: every rank's input slices uploaded once; extract_resident() then starts from HBM."""
        vol = _f32(vol)
        assert vol.shape == self.shape, (vol.shape, self.shape)
        err = C.create_string_buffer(512)
        rc = self._L.sift3d_zslab_set_volume(self._h, vol.ctypes.data, err, 512)
        if rc != 0:
            e = Sift3DError("sift3d_zslab_set_volume -> %d: %s" % (rc, err.value.decode(errors="replace")))
            e.code = rc
            raise e

    def extract_resident(self, initial_image_scale=1.0, desc_mode=DESC_SIFT, eig_thres=140.0, size_factor=1.0, copy=True):
        """sift3d_zslab_extract_resident: (records, stats).  copy=False: a view of the handle's merge buffer, valid until
        the handle's next call."""
        view, n, st, err = C.c_void_p(), C.c_int64(0), ZSlabStats(), C.create_string_buffer(512)
        rc = self._L.sift3d_zslab_extract_resident(self._h, float(initial_image_scale), int(desc_mode), float(eig_thres), float(size_factor),
                                                   C.byref(view), C.byref(n), C.byref(st), err, 512)
        if rc != 0:
            e = Sift3DError("sift3d_zslab_extract_resident -> %d: %s" % (rc, err.value.decode(errors="replace")))
            e.code = rc
            raise e
        if n.value:
            recs = np.frombuffer((C.c_char * (n.value * FEATURE_DTYPE.itemsize)).from_address(view.value), FEATURE_DTYPE, n.value)
            recs = recs.copy() if copy else recs
        else:
            recs = np.zeros(0, FEATURE_DTYPE)
        return recs, {k: getattr(st, k) for k, _ in ZSlabStats._fields_}

    def set_tuning(self, knob, value):
        rc = self._L.sift3d_zslab_set_tuning(self._h, int(knob), int(value))
        if rc != 0:
            raise Sift3DError("sift3d_zslab_set_tuning(%d, %d) -> %d" % (knob, value, rc))

    def close(self):
        if self._h:
            self._L.sift3d_zslab_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


GROUPS = 193   # include/sift3d.h: SIFT3D_GROUPS


def host_register(address, nbytes):
    """hipHostRegister (portable, mapped) on the caller's pages: every device of this process may store into them."""
    rc = hip_lib().sift3d_host_register(C.c_void_p(int(address)), int(nbytes))
    if rc != 0:
        raise Sift3DError("sift3d_host_register failed (%d)" % rc)


def host_unregister(address):
    hip_lib().sift3d_host_unregister(C.c_void_p(int(address)))


def device_count():
    return int(hip_lib().sift3d_device_count())


# ---- matcher: exact nearest neighbours on the GPU, votes on the host ---------------------------
FEATMATCH = os.path.join(CSRC, "_build", "featMatchMultiple")


def knn64(db, queries, k, device=0, repeats=1):
    """sift3d_knn64: for every row of `queries` (n_q x 64 int8, components 0..127) the k nearest rows of `db`, ascending
    by (squared distance, index).  Returns (idx, dist2, kernel_ms), idx / dist2 of shape (n_q, k)."""
    db = np.ascontiguousarray(db, np.int8)
    queries = np.ascontiguousarray(queries, np.int8)
    assert db.ndim == 2 and db.shape[1] == 64 and queries.ndim == 2 and queries.shape[1] == 64
    idx = np.empty((len(queries), k), np.int32)
    d2 = np.empty((len(queries), k), np.int32)
    ms, err = C.c_double(0.0), C.create_string_buffer(256)
    rc = hip_lib().sift3d_knn64(int(device), db.ctypes.data, len(db), queries.ctypes.data, len(queries), int(k), idx.ctypes.data,
                                d2.ctypes.data, int(repeats), C.byref(ms), err, 256)
    if rc != 0:
        e = Sift3DError("sift3d_knn64 -> %d: %s" % (rc, err.value.decode(errors="replace")))
        e.code = rc
        raise e
    return idx, d2, ms.value


def match_filter(feats, reoriented=1, peaks=4):
    """sift3d_match_filter: the reference matcher's feature filters; returns the kept records."""
    f = np.ascontiguousarray(feats, FEATURE_DTYPE).copy()
    n = host_lib().sift3d_match_filter(f.ctypes.data, len(f), int(reoriented), int(peaks))
    return f[:n].copy()


def match_descriptors(feats):
    f = np.ascontiguousarray(feats, FEATURE_DTYPE)
    out = np.empty((len(f), 64), np.int8)
    if host_lib().sift3d_match_descriptors(f.ctypes.data, len(f), out.ctypes.data) != 0:
        raise Sift3DError("a descriptor value is outside 0..127")
    return out


def match_votes(first, labels, n_labels, nn_idx, nn_dist2):
    """sift3d_match_votes: (votes, counts), each n_images x n_labels."""
    first = np.ascontiguousarray(first, np.int64)
    labels = np.ascontiguousarray(labels, np.int32)
    nn_idx = np.ascontiguousarray(nn_idx, np.int32)
    nn_dist2 = np.ascontiguousarray(nn_dist2, np.int32)
    n_img, k = len(first) - 1, nn_idx.shape[1]
    votes = np.zeros((n_img, n_labels), np.float32)
    counts = np.zeros((n_img, n_labels), np.int32)
    rc = host_lib().sift3d_match_votes(None, first.ctypes.data, n_img, labels.ctypes.data, int(n_labels), nn_idx.ctypes.data,
                                       nn_dist2.ctypes.data, k, votes.ctypes.data, counts.ctypes.data)
    if rc != 0:
        raise Sift3DError("sift3d_match_votes -> %d" % rc)
    return votes, counts


LIBM_CURRENT, LIBM_GCC5 = 0, 1   # sift3d_set_libm_variant


def set_libm_variant(which):
    """Process-wide: which build of the reference the Gaussian taps follow (include/sift3d.h); returns the previous setting."""
    r = hip_lib().sift3d_set_libm_variant(int(which))
    if r < 0:
        raise Sift3DError("sift3d_set_libm_variant(%r) -> %d" % (which, r))
    return r


def gauss_taps(sigma, min_value=0.01):
    t = np.zeros(129, np.float32)
    n = hip_lib().sift3d_gauss_taps(float(sigma), float(min_value), t.ctypes.data)
    if n < 0:
        raise Sift3DError("sift3d_gauss_taps(%r, %r) -> %d" % (sigma, min_value, n))
    return t[:n].copy()


def synth_blobs(nx, ny, nz, seed=12345):
    """Deterministic blob-field volume, shape (nz, ny, nx) float32."""
    v = np.empty((nz, ny, nx), np.float32)
    host_lib().sift3d_synth_blobs(v.ctypes.data, nx, ny, nz, seed)
    return v


def synth_blobs_slices(nx, ny, nz, z0, z1, seed=12345):
    """Planes [z0, z1) of synth_blobs(nx, ny, nz, seed), without making the rest: shape (z1 - z0, ny, nx)."""
    z0, z1 = max(0, int(z0)), min(int(nz), int(z1))
    v = np.empty((max(0, z1 - z0), ny, nx), np.float32)
    if z1 > z0:
        host_lib().sift3d_synth_blobs_slices(v.ctypes.data, nx, ny, nz, seed, z0, z1)
    return v


class _NiftiMinImage(C.Structure):
    """nifti_min_image (csrc/nifti_min.h)"""
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("nt", C.c_int), ("dx", C.c_float), ("dy", C.c_float),
                ("dz", C.c_float), ("datatype", C.c_int), ("qform_code", C.c_int), ("sform_code", C.c_int),
                ("qto_xyz", C.c_float * 16), ("sto_xyz", C.c_float * 16), ("data", C.POINTER(C.c_float))]


def nifti_fast_inflate(on):
    """csrc/nifti_min.h: 0 = gzip'ed files through zlib only; 1 (default) = through libdeflate where the system has it."""
    host_lib().nifti_min_fast_inflate(int(bool(on)))


def nifti_fast_inflate_count():
    return int(host_lib().nifti_min_fast_inflate_count())


def read_nifti(path):
    """nifti_min_read: (volume float32 of shape (nt*nz, ny, nx), header dict).  Raises Sift3DError with the reader's code."""
    img = _NiftiMinImage()
    rc = host_lib().nifti_min_read(os.fsencode(path), C.byref(img))
    if rc != 0:
        e = Sift3DError("nifti_min_read(%s) -> %d" % (path, rc))
        e.code = rc
        raise e
    try:
        n = img.nx * img.ny * img.nz * img.nt
        vol = np.ctypeslib.as_array(img.data, shape=(n,)).copy().reshape(img.nt * img.nz, img.ny, img.nx)
        hdr = {"dims": (img.nx, img.ny, img.nz, img.nt), "voxel": (img.dx, img.dy, img.dz), "datatype": img.datatype,
               "qform_code": img.qform_code, "sform_code": img.sform_code,
               "qto_xyz": np.array(img.qto_xyz, np.float32).reshape(4, 4), "sto_xyz": np.array(img.sto_xyz, np.float32).reshape(4, 4)}
    finally:
        host_lib().nifti_min_free(C.byref(img))
    return vol, hdr


def write_nifti(path, vol, voxel=(1.0, 1.0, 1.0), qform=None, sform=None):
    """float32 .nii writer.  qform = (quatern_b, c, d, qoffset_x, y, z, qfac), sform = 12 floats (srow_x, y, z)."""
    vol = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = vol.shape
    q = None if qform is None else np.ascontiguousarray(qform, np.float32).reshape(7)
    s = None if sform is None else np.ascontiguousarray(sform, np.float32).reshape(12)
    rc = host_lib().nifti_min_write_f32_ex(os.fsencode(path), vol.ctypes.data, nx, ny, nz, *[float(v) for v in voxel],
                                           None if q is None else q.ctypes.data, None if s is None else s.ctypes.data)
    if rc != 0:
        raise Sift3DError("could not write %s" % path)


def write_key(path, feats, eig_thres=140.0, comments=()):
    feats = np.ascontiguousarray(feats, FEATURE_DTYPE)
    arr = (C.c_char_p * max(1, len(comments)))(*[c.encode() for c in comments])
    rc = host_lib().sift3d_write_key(os.fsencode(path), feats.ctypes.data, len(feats), float(eig_thres), len(comments),
                                     C.cast(arr, C.c_void_p))
    if rc != 0:
        raise Sift3DError("could not write %s" % path)


def write_key_bin(path, feats, eig_thres=140.0):
    """msFeature3DVectorOutputBin: header lines as text, then fixed-size binary records."""
    feats = np.ascontiguousarray(feats, FEATURE_DTYPE)
    if host_lib().sift3d_write_key_bin(os.fsencode(path), feats.ctypes.data, len(feats), float(eig_thres)) != 0:
        raise Sift3DError("could not write %s" % path)


def world_transform(feats, m44):
    """sift3d_world_transform (featExtract.cpp:436-538): records to world coordinates through a 4 x 4 voxel-to-mm matrix."""
    out = np.ascontiguousarray(feats, FEATURE_DTYPE).copy()
    m = np.ascontiguousarray(m44, np.float32).reshape(4, 4)
    host_lib().sift3d_world_transform(out.ctypes.data, len(out), m.ctypes.data)
    return out


def write_pgm(path, slice_yx):
    """output_float + GenericImage::WriteToFile: an x-y slice of floats as the reference's image.pgm."""
    a = _f32(slice_yx)
    if a.ndim != 2 or host_lib().sift3d_write_pgm(os.fsencode(path), a.ctypes.data, a.shape[0], a.shape[1]) != 0:
        raise Sift3DError("could not write %s" % path)


def read_key(path):
    """msFeature3DVectorInputText: the records of a text .key file as a FEATURE_DTYPE array."""
    L = host_lib()
    ptr, n = C.c_void_p(), C.c_int64(0)
    rc = L.sift3d_read_key(os.fsencode(path), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise Sift3DError("could not read %s (%d)" % (path, rc))
    out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n.value * FEATURE_DTYPE.itemsize,)).view(FEATURE_DTYPE).copy()
    L.free_ptr(ptr)
    return out


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Context:
    """Device-resident pyramid for volumes of up to nx*ny*nz voxels on one HIP device."""

    def __init__(self, nx, ny, nz, device=0, slab=False):
        """slab=True: sift3d_create_slab -- a context for the Z-slab building blocks, without level buffers of its own."""
        self._L = hip_lib()
        if self._L.sift3d_device_count() <= 0:
            raise Sift3DError("no HIP device visible and there is no CPU fallback")
        make = self._L.sift3d_create_slab if slab else self._L.sift3d_create
        self._h = make(int(device), int(nx), int(ny), int(nz))
        if not self._h:
            raise Sift3DError("sift3d_create%s(device=%d, %d x %d x %d) failed" % ("_slab" if slab else "", device, nx, ny, nz))
        self.device = device

    def set_tuning(self, knob, value):
        """sift3d_set_tuning: TUNE_* knobs (tests and A/B timing; no knob changes a result)."""
        self._chk(self._L.sift3d_set_tuning(self._h, int(knob), int(value)), "sift3d_set_tuning")

    def host_buffer_grows(self):
        """sift3d_host_buffer_grows: runs on this context that outgrew their pinned record buffers."""
        return int(self._L.sift3d_host_buffer_grows(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.sift3d_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _chk(self, rc, what):
        if rc != 0:
            raise Sift3DError("%s -> %d: %s" % (what, rc, self._L.sift3d_last_error(self._h).decode()))

    # ---- operator level (host arrays shaped (nz, ny, nx)) ----
    def gauss_blur(self, vol, sigma, min_value=0.01):
        vol = _f32(vol)
        nz, ny, nx = vol.shape
        out = np.empty_like(vol)
        self._chk(self._L.sift3d_gauss_blur(self._h, vol.ctypes.data, out.ctypes.data, nx, ny, nz, float(sigma),
                                            float(min_value)), "sift3d_gauss_blur")
        return out

    def dog(self, a, b):
        a, b = _f32(a), _f32(b)
        out = np.empty_like(a)
        self._chk(self._L.sift3d_dog(self._h, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size), "sift3d_dog")
        return out

    def subsample2(self, vol):
        vol = _f32(vol)
        nz, ny, nx = vol.shape
        out = np.empty((nz // 2, ny // 2, nx // 2), np.float32)
        self._chk(self._L.sift3d_subsample2(self._h, vol.ctypes.data, nx, ny, nz, out.ctypes.data), "sift3d_subsample2")
        return out

    def double_size(self, vol):
        vol = _f32(vol)
        nz, ny, nx = vol.shape
        out = np.empty((2 * nz, 2 * ny, 2 * nx), np.float32)
        self._chk(self._L.sift3d_double_size(self._h, vol.ctypes.data, nx, ny, nz, out.ctypes.data), "sift3d_double_size")
        return out

    def halve_size(self, vol):
        vol = _f32(vol)
        nz, ny, nx = vol.shape
        out = np.empty((nz // 2, ny // 2, nx // 2), np.float32)
        self._chk(self._L.sift3d_halve_size(self._h, vol.ctypes.data, nx, ny, nz, out.ctypes.data), "sift3d_halve_size")
        return out

    def selftest_lds_add(self, a, b):
        """(a + b on the vector ALU, a + b through ds_add_f32), both computed on the device."""
        a, b = _f32(a).ravel(), _f32(b).ravel()
        valu, lds = np.empty_like(a), np.empty_like(a)
        self._chk(self._L.sift3d_selftest_lds_add(self._h, a.ctypes.data, b.ctypes.data, a.size, valu.ctypes.data,
                                                  lds.ctypes.data), "sift3d_selftest_lds_add")
        return valu, lds

    def extrema(self, d_prev, d_cur, d_next=None, capacity=None):
        """Returns (minima, maxima) structured arrays in raster order."""
        d_prev, d_cur = _f32(d_prev), _f32(d_cur)
        nz, ny, nx = d_cur.shape
        d_next = None if d_next is None else _f32(d_next)
        cap = int(capacity) if capacity is not None else d_cur.size // 8 + 1024
        mins = np.zeros(cap, EXTREMUM_DTYPE)
        maxs = np.zeros(cap, EXTREMUM_DTYPE)
        nmin, nmax = C.c_int64(0), C.c_int64(0)
        rc = self._L.sift3d_extrema(self._h, d_prev.ctypes.data, d_cur.ctypes.data,
                                    None if d_next is None else d_next.ctypes.data, nx, ny, nz, mins.ctypes.data, cap,
                                    C.byref(nmin), maxs.ctypes.data, cap, C.byref(nmax))
        self._chk(rc, "sift3d_extrema")
        return mins[:nmin.value].copy(), maxs[:nmax.value].copy()

    # ---- pipeline level ----
    def set_volume(self, vol, resize=0):
        """resize: +1 / -1 = the -2+ / -2- options (doubled / halved on the device after the upload)."""
        vol = _f32(vol)
        nz, ny, nx = vol.shape
        self._chk(self._L.sift3d_set_volume_resized(self._h, vol.ctypes.data, nx, ny, nz, int(resize)), "sift3d_set_volume_resized")

    def reserve(self, n_extrema):
        self._chk(self._L.sift3d_reserve(self._h, int(n_extrema)), "sift3d_reserve")

    def set_volume_in_runs(self, vol, runs, resize=0):
        """sift3d_set_volume_begin / _planes / _end: the volume handed over in the given runs of planes [(z0, n), ...]."""
        vol = _f32(vol)
        nz, ny, nx = vol.shape
        self._chk(self._L.sift3d_set_volume_begin(self._h, nx, ny, nz, int(resize)), "sift3d_set_volume_begin")
        for z0, n in runs:
            part = vol[z0:z0 + n]
            self._chk(self._L.sift3d_set_volume_planes(self._h, part.ctypes.data, int(z0), int(n)), "sift3d_set_volume_planes")
        self._chk(self._L.sift3d_set_volume_end(self._h), "sift3d_set_volume_end")

    def set_volume_dev(self, dev_ptr, nx, ny, nz):
        self._chk(self._L.sift3d_set_volume_dev(self._h, C.c_void_p(int(dev_ptr)), nx, ny, nz), "sift3d_set_volume_dev")

    def detect(self, initial_image_scale=1.0):
        out, n = C.c_void_p(), C.c_int64(0)
        self._chk(self._L.sift3d_detect(self._h, float(initial_image_scale), C.byref(out), C.byref(n)), "sift3d_detect")
        try:
            buf = (C.c_char * (n.value * CANDIDATE_DTYPE.itemsize)).from_address(out.value) if n.value else b""
            return np.frombuffer(buf, CANDIDATE_DTYPE, n.value).copy()
        finally:
            self._L.sift3d_free(out)

    def extract(self, initial_image_scale=1.0, desc_mode=DESC_SIFT, eig_thres=140.0, size_factor=1.0, copy=True):
        """Full extraction -> structured array of records.  copy=False returns a view of the context's
        pinned download buffer, valid until the next call on this context."""
        out, n = C.c_void_p(), C.c_int64(0)
        self._chk(self._L.sift3d_extract_view(self._h, float(initial_image_scale), int(desc_mode), float(eig_thres),
                                              float(size_factor), C.byref(out), C.byref(n)), "sift3d_extract_view")
        if n.value == 0:
            return np.zeros(0, FEATURE_DTYPE)
        buf = (C.c_char * (n.value * FEATURE_DTYPE.itemsize)).from_address(out.value)
        a = np.frombuffer(buf, FEATURE_DTYPE, n.value)
        return a.copy() if copy else a

    # ---- device-pointer forms (bench) ----
    def gauss_blur_dog_dev(self, d_in, d_out, d_dog, nx, ny, nz, sigma, min_value=0.01):
        self._chk(self._L.sift3d_gauss_blur_dog_dev(self._h, C.c_void_p(int(d_in)), C.c_void_p(int(d_out)) if d_out else None,
                                                    C.c_void_p(int(d_dog)) if d_dog else None, nx, ny, nz, float(sigma),
                                                    float(min_value)), "sift3d_gauss_blur_dog_dev")

    def gauss_blur_dog_half_dev(self, d_in, d_out, d_dog, d_half, nx, ny, nz, sigma, min_value=0.01):
        """sift3d_gauss_blur_dog_half_dev: level, DoG and the half-size volume; returns True when one launch made all three."""
        one = C.c_int(0)
        self._chk(self._L.sift3d_gauss_blur_dog_half_dev(self._h, C.c_void_p(int(d_in)), C.c_void_p(int(d_out)),
                                                         C.c_void_p(int(d_dog)) if d_dog else None, C.c_void_p(int(d_half)), nx, ny, nz,
                                                         float(sigma), float(min_value), C.byref(one)), "sift3d_gauss_blur_dog_half_dev")
        return bool(one.value)

    def gauss_blur_dev(self, d_in, d_out, nx, ny, nz, sigma, min_value=0.01):
        self._chk(self._L.sift3d_gauss_blur_dev(self._h, C.c_void_p(int(d_in)), C.c_void_p(int(d_out)), nx, ny, nz,
                                                float(sigma), float(min_value)), "sift3d_gauss_blur_dev")

    def blur_window_supported(self, nx, ny, sigma, min_value=0.01):
        return bool(self._L.sift3d_blur_window_supported(nx, ny, float(sigma), float(min_value)))

    def gauss_blur_dog_window_dev(self, d_in, d_out, d_dog, nx, ny, nz, z_lo, z_hi, sigma, min_value=0.01):
        """sift3d_gauss_blur_dog_window_dev: only the output planes [z_lo, z_hi) are produced."""
        self._chk(self._L.sift3d_gauss_blur_dog_window_dev(self._h, C.c_void_p(int(d_in)), C.c_void_p(int(d_out)) if d_out else None,
                                                           C.c_void_p(int(d_dog)) if d_dog else None, nx, ny, nz, int(z_lo), int(z_hi),
                                                           float(sigma), float(min_value)), "sift3d_gauss_blur_dog_window_dev")

    def dog_dev(self, d_a, d_b, d_out, n):
        self._chk(self._L.sift3d_dog_dev(self._h, C.c_void_p(int(d_a)), C.c_void_p(int(d_b)), C.c_void_p(int(d_out)), n),
                  "sift3d_dog_dev")

    def subsample2_dev(self, d_in, nx, ny, nz, d_out):
        self._chk(self._L.sift3d_subsample2_dev(self._h, C.c_void_p(int(d_in)), nx, ny, nz, C.c_void_p(int(d_out))),
                  "sift3d_subsample2_dev")

    # ---- Z-slab building blocks ----
    def candidates_reset(self):
        self._chk(self._L.sift3d_candidates_reset(self._h), "sift3d_candidates_reset")

    def extrema_append_dev(self, d_prev, d_cur, d_next, nx, ny, nz_local, level_id, z_lo, z_hi):
        self._chk(self._L.sift3d_extrema_append_dev(self._h, C.c_void_p(int(d_prev)), C.c_void_p(int(d_cur)),
                                                    C.c_void_p(int(d_next)), nx, ny, nz_local, int(level_id), int(z_lo),
                                                    int(z_hi)), "sift3d_extrema_append_dev")

    def lazy_levels_supported(self, nx, ny, nz_local, next_sigma):
        return bool(self._L.sift3d_lazy_levels_supported(nx, ny, nz_local, float(next_sigma)))

    def extrema_append_lazy_dev(self, d_prev, g_prev_a, g_prev_b, d_cur, d_next, g_next, next_sigma, nx, ny, nz_local, level_id,
                                z_lo, z_hi):
        """Like extrema_append_dev with a neighbour level given as Gaussian levels instead of a stored DoG volume (0 = not
        given): the level below as g_prev_a - g_prev_b, the level above as g_next - blur(g_next, next_sigma)."""
        vp = lambda v: C.c_void_p(int(v)) if v else None
        self._chk(self._L.sift3d_extrema_append_lazy_dev(self._h, vp(d_prev), vp(g_prev_a), vp(g_prev_b), vp(d_cur), vp(d_next),
                                                         vp(g_next), float(next_sigma), nx, ny, nz_local, int(level_id), int(z_lo),
                                                         int(z_hi)), "sift3d_extrema_append_lazy_dev")

    @staticmethod
    def _level_array(levels):
        arr = (LevelDesc * len(levels))()
        for i, lv in enumerate(levels):
            arr[i] = LevelDesc(int(lv["img"]), int(lv["dogc"]), lv["nx"], lv["ny"], lv["nz_local"], lv["nz_global"],
                               lv["z_offset"], lv["sigma_h"], lv["sigma_c"], lv["sigma_l"], lv["octave_factor"])
        return arr

    def candidates_dev(self, levels):
        arr = self._level_array(levels)
        out, n = C.c_void_p(), C.c_int64(0)
        self._chk(self._L.sift3d_candidates_dev(self._h, arr, len(levels), C.byref(out), C.byref(n)), "sift3d_candidates_dev")
        try:
            buf = (C.c_char * (n.value * CANDIDATE_DTYPE.itemsize)).from_address(out.value) if n.value else b""
            return np.frombuffer(buf, CANDIDATE_DTYPE, n.value).copy()
        finally:
            self._L.sift3d_free(out)

    def describe_dev(self, levels, desc_mode=DESC_SIFT, eig_thres=140.0, size_factor=1.0, copy=True):
        """Returns (records, group); group = level_id*2 + is_max per record.  copy=False returns views of the
        context's pinned download buffers, valid until the next call on this context."""
        arr = self._level_array(levels)
        view, grp, n = C.c_void_p(), C.c_void_p(), C.c_int64(0)
        self._chk(self._L.sift3d_describe_dev(self._h, arr, len(levels), int(desc_mode), float(eig_thres), float(size_factor),
                                              C.byref(view), C.byref(grp), C.byref(n)), "sift3d_describe_dev")
        if n.value == 0:
            return np.zeros(0, FEATURE_DTYPE), np.zeros(0, np.int32)
        rb = (C.c_char * (n.value * FEATURE_DTYPE.itemsize)).from_address(view.value)
        gb = (C.c_char * (n.value * 4)).from_address(grp.value)
        recs, grp = np.frombuffer(rb, FEATURE_DTYPE, n.value), np.frombuffer(gb, np.int32, n.value)
        return (recs.copy(), grp.copy()) if copy else (recs, grp)

    def describe_dev_counts(self, levels, desc_mode=DESC_SIFT, eig_thres=140.0, size_factor=1.0):
        """First half of describe_dev for a caller that places several contexts' records in one list (include/sift3d.h): this
        context's records per group (GROUPS int32, a copy) and their sum.  describe_dev_place must follow."""
        arr = self._level_array(levels)
        cnt, n = C.c_void_p(), C.c_int64(0)
        self._chk(self._L.sift3d_describe_dev_counts(self._h, arr, len(levels), int(desc_mode), float(eig_thres), float(size_factor),
                                                     C.byref(cnt), C.byref(n)), "sift3d_describe_dev_counts")
        return np.frombuffer((C.c_char * (GROUPS * 4)).from_address(cnt.value), np.int32, GROUPS).copy(), n.value

    def describe_dev_place(self, list_address, shift):
        """Second half: the descriptor kernel stores record i of group g at list[i + shift[g]] (list_address: registered host memory,
        host_register).  Returns this context's record count -- or, with list_address None (the list turned out too small), what
        describe_dev(copy=False) returns: views of the context's own buffers."""
        n, own, grp = C.c_int64(0), C.c_void_p(), C.c_void_p()
        sh = np.ascontiguousarray(shift, np.int32) if shift is not None else None
        self._chk(self._L.sift3d_describe_dev_place(self._h, C.c_void_p(int(list_address)) if list_address else None,
                                                    sh.ctypes.data_as(C.c_void_p) if sh is not None else None, C.byref(own), C.byref(grp),
                                                    C.byref(n)), "sift3d_describe_dev_place")
        if list_address:
            return n.value
        if n.value == 0:
            return np.zeros(0, FEATURE_DTYPE), np.zeros(0, np.int32)
        rb = (C.c_char * (n.value * FEATURE_DTYPE.itemsize)).from_address(own.value)
        gb = (C.c_char * (n.value * 4)).from_address(grp.value)
        return np.frombuffer(rb, FEATURE_DTYPE, n.value), np.frombuffer(gb, np.int32, n.value)

    def set_max_octaves(self, n):
        """0 = the reference's stop rule (default); n > 0 = at most n octaves."""
        self._chk(self._L.sift3d_set_max_octaves(self._h, int(n)), "sift3d_set_max_octaves")

    def sync(self):
        self._chk(self._L.sift3d_sync(self._h), "sift3d_sync")

    def set_stream(self, hip_stream):
        self._chk(self._L.sift3d_set_stream(self._h, C.c_void_p(int(hip_stream)) if hip_stream else None), "sift3d_set_stream")

    def enable_timing(self, on=True):
        """False / 0 off, True / 1 every launch, 2 only the blur launches on the full-size volume, 3 every launch with the
        extrema kept on the main stream (each launch timed alone)."""
        self._chk(self._L.sift3d_enable_timing(self._h, int(on)), "sift3d_enable_timing")

    def launch_log(self):
        """Per-launch records (stage, ntaps, nvox, alg_bytes, ms) of the last pipeline/blur call."""
        n = C.c_int64(0)
        self._L.sift3d_get_launch_log(self._h, None, 0, C.byref(n))
        out = np.zeros(max(1, n.value), LAUNCH_DTYPE)
        self._chk(self._L.sift3d_get_launch_log(self._h, out.ctypes.data, len(out), C.byref(n)), "sift3d_get_launch_log")
        return out[:n.value]

    def timings(self):
        t = _Timings()
        self._chk(self._L.sift3d_get_timings(self._h, C.byref(t)), "sift3d_get_timings")
        d = {"total_ms": t.total_ms, "n_octaves": t.n_octaves, "n_extrema": t.n_extrema,
             "n_keypoints": t.n_keypoints, "n_records": t.n_records, "stages": {}}
        for i, s in enumerate(STAGES):
            d["stages"][s] = {"ms": t.ms[i], "launches": t.launches[i], "alg_bytes": t.alg_bytes[i]}
        return d
