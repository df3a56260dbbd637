"""CPU suite: the C-ABI surface and the host-side helpers (no compute calls)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from keyio import read_key

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sift3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sift3d_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("sift3d_device_count", "sift3d_create", "sift3d_destroy", "sift3d_gauss_blur", "sift3d_dog",
                 "sift3d_subsample2", "sift3d_extrema", "sift3d_extract", "sift3d_detect", "sift3d_free"):
        assert must in names


def test_library_exports_every_declared_symbol(built):
    lib = C.CDLL(built.LIB_HIP)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_library_has_gfx950_code_object(built):
    out = subprocess.run(["strings", "-n", "6", built.LIB_HIP], capture_output=True, text=True).stdout
    assert "gfx950" in out
    # no CPU fallback hides in the product: the oracle is not linked
    assert "o3_extract" not in out and "libsift3d_oracle" not in out


def test_product_sources_do_not_reference_the_oracle():
    bad = []
    for d, _, fs in os.walk(os.path.join(ROOT, "3d_sift_cuda_amd")):
        if "_build" in d:
            continue
        for f in fs:
            if f.endswith((".c", ".h", ".hip", ".py", "Makefile")):
                t = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"sift3d_oracle|oracle/|import _oracle|o3_[a-z]+\(", t):
                    bad.append(f)
    assert not bad, bad


def test_gauss_taps_host(built, oracle):
    for s in (0.5, 0.95, 1.2262736558914185, 1.5198684930801392, 3.0900158882141113, 0.0):
        a = built.gauss_taps(s)
        b = oracle.taps(s)
        assert len(a) == len(b) and (a.view(np.uint32) == b.view(np.uint32)).all()


def test_no_gpu_means_loud_failure(built):
    if built.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(built.Sift3DError):
        built.Context(8, 8, 8)


def test_synth_is_deterministic(built):
    a = built.synth_blobs(40, 32, 24, seed=9)
    b = built.synth_blobs(40, 32, 24, seed=9)
    c = built.synth_blobs(40, 32, 24, seed=10)
    assert a.shape == (24, 32, 40) and (a == b).all() and not (a == c).all()
    assert np.isfinite(a).all() and a.std() > 1.0


def test_nifti_roundtrip_and_cli_usage(built, tmp_path, oracle):
    import _oracle
    vol = built.synth_blobs(20, 18, 16, seed=3)
    for name in ("v.nii", "v.nii.gz"):
        p = str(tmp_path / name)
        built.write_nifti(p, vol, voxel=(1.0, 2.0, 3.0))
        k = str(tmp_path / (name + ".key"))
        r = subprocess.run([_oracle.CLI, p, k], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Input image: i=20 j=18 k=16" in r.stdout
        head = open(k).read().splitlines()
        assert head[0] == "# featExtract 1.1"
        assert head[2] == "# Extraction Voxel Size (mm)  (ijk) : 1.000000 2.000000 3.000000"
    # the product CLI: usage text and exit code of featExtract.cpp:285-289,344-348
    r = subprocess.run([built.FEATEXTRACT], capture_output=True, text=True)
    assert r.returncode == 255 and "Usage: featExtract [options] <input image> <output features>" in r.stdout
    r = subprocess.run([built.FEATEXTRACT, "-q", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 255 and "Error: unknown command line argument: -q" in r.stdout


def test_key_writer_matches_oracle_writer(built, oracle, tmp_path):
    vol = built.synth_blobs(64, 64, 64)
    recs, _ = oracle.extract(vol)
    cm = ["a", "b", "c"]
    p1, p2 = str(tmp_path / "a.key"), str(tmp_path / "b.key")
    oracle.write_key(p1, recs, comments=cm)
    built.write_key(p2, recs.astype(built.FEATURE_DTYPE), comments=cm)
    assert open(p1, "rb").read() == open(p2, "rb").read()


def test_key_writer_number_formatting(built, oracle, tmp_path):
    """The product's writer formats "%f" / "%d" itself (keyfile.c); the oracle's uses fprintf.  Random bit patterns,
    ties at the sixth decimal, signed zeros, denormals, huge values, infinities and NaNs must give the same bytes."""
    rng = np.random.default_rng(3)
    n = 4000
    recs = np.zeros(n, built.FEATURE_DTYPE)
    special = np.array([0.0, -0.0, 0.0078125, -0.0078125, 0.5 ** 7 * 3, 1e-7, -4.9e-7, 5e-7, 0.9999995, 0.99999949, 1.5e-45,
                        -1.5e-45, 3.4e38, -3.4e38, 4.0e9, 3.9999998e9, 4.1e9, np.inf, -np.inf, np.nan, 123456.789, 2.0 ** 31,
                        16777216.0, 0.1, 0.2, 0.3], np.float32)
    for f, k in (("x", 1), ("y", 1), ("z", 1), ("scale", 1), ("ori", 9), ("eigs", 3)):
        bits = rng.integers(0, 1 << 32, n * k, dtype=np.uint64).astype(np.uint32)
        vals = bits.view(np.float32).copy()
        pick = rng.random(n * k) < 0.5                     # half of them ordinary magnitudes
        vals[pick] = ((rng.random(pick.sum()) - 0.5) * 10.0 ** rng.integers(-7, 8, pick.sum())).astype(np.float32)
        tie = rng.random(n * k) < 0.1                      # exact ties: odd multiples of 2^-7 scaled by powers of two
        vals[tie] = ((2 * rng.integers(0, 5000, tie.sum()) + 1) * 0.5 ** rng.integers(7, 12, tie.sum())).astype(np.float32)
        sp = rng.random(n * k) < 0.05
        vals[sp] = special[rng.integers(0, len(special), sp.sum())]
        recs[f] = vals.reshape(recs[f].shape)
    recs["info"] = rng.integers(0, 1 << 31, n).astype(np.uint32)
    recs["desc"] = rng.integers(-128, 128, (n, 64)).astype(np.float32)
    p1, p2 = str(tmp_path / "a.key"), str(tmp_path / "b.key")
    oracle.write_key(p1, recs, eig_thres=-1.0, comments=["a", "b", "c"])
    built.write_key(p2, recs, eig_thres=-1.0, comments=["a", "b", "c"])
    a, b = open(p1, "rb").read(), open(p2, "rb").read()
    assert a.count(b"\n") == n + 6
    assert a == b


def test_key_reader_and_binary_writer(built, oracle, tmp_path):
    """msFeature3DVectorInputText / msFeature3DVectorOutputBin (MultiScale.h:228-384) on the host side."""
    import struct
    gold = os.path.join(ROOT, "tests", "golden", "refbin_blob64.key")   # written by the reference's shipped binary
    recs = built.read_key(gold)
    ref = read_key(gold)
    assert len(recs) == ref["count"] == 74
    assert np.allclose(recs["x"], ref["rows"][:, 0]) and np.allclose(recs["scale"], ref["rows"][:, 3])
    assert (recs["ori"] == ref["rows"][:, 4:13].astype(np.float32)).all()
    assert (recs["info"] == ref["rows"][:, 16].astype(np.uint32)).all()
    assert (recs["desc"] == ref["rows"][:, 17:].astype(np.float32)).all()
    # text round trip: what the reader returns writes back to the same file body
    p = str(tmp_path / "again.key")
    built.write_key(p, recs, comments=[h[2:] for h in ref["header"][1:4]])
    assert open(p).read() == open(gold).read()
    # binary flavour: two text header lines, then 4+9+3 floats, the info word and 64 unsigned chars per record
    b = str(tmp_path / "out.bin")
    built.write_key_bin(b, recs)
    raw = open(b, "rb").read()
    head = b"# featExtract 1.1\nFeatures: 74\n"
    assert raw.startswith(head) and len(raw) == len(head) + 74 * (16 * 4 + 4 + 64)
    first = struct.unpack_from("<16fI64B", raw, len(head))
    assert np.float32(first[0]) == recs["x"][0] and first[16] == recs["info"][0]
    assert list(first[17:]) == [int(v) for v in recs["desc"][0]]
    # errors: missing file, no header
    with pytest.raises(built.Sift3DError):
        built.read_key(str(tmp_path / "missing.key"))
    bad = str(tmp_path / "bad.key")
    open(bad, "w").write("# featExtract 1.1\nnothing here\n")
    with pytest.raises(built.Sift3DError):
        built.read_key(bad)
