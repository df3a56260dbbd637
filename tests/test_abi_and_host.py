"""CPU suite: the C-ABI surface and the host-side helpers (no compute calls)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from keyio import read_key

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="sift3d.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef SIFT3D_DEV.*?#endif", "", src, flags=re.S)   # development builds only (make DEV=1), not the product
    return sorted(set(re.findall(r"\b(sift3d_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("sift3d_device_count", "sift3d_create", "sift3d_destroy", "sift3d_gauss_blur", "sift3d_dog",
                 "sift3d_subsample2", "sift3d_extrema", "sift3d_extract", "sift3d_detect", "sift3d_free"):
        assert must in names


def test_library_exports_every_declared_symbol(built):
    lib = C.CDLL(built.LIB_HIP)
    missing = [n for n in declared_functions() + declared_functions("sift3d_dev.h") if not hasattr(lib, n)]
    assert not missing, missing
    # include/sift3d.h is the boundary and nothing else (round-5 review): development hooks and the self-test are declared in
    # include/sift3d_dev.h; of those the product library exports the self-test only
    assert not [n for n in declared_functions() if n.startswith("sift3d_dev_") or "selftest" in n]
    assert declared_functions("sift3d_dev.h") == ["sift3d_selftest_lds_add"]
    dev = open(os.path.join(ROOT, "include", "sift3d_dev.h")).read()
    for hook in re.findall(r"\b(sift3d_dev_[a-z0-9_]+)\s*\(", dev):
        assert not hasattr(lib, hook), hook


def test_python_tuning_constants_mirror_the_header(built):
    """The TUNE_* constants of the Python host side are the header's sift3d_tuning enum, name by name and in its order:
    a knob added to one side only would silently set another knob."""
    src = open(os.path.join(ROOT, "include", "sift3d.h")).read()
    end = src.index("} sift3d_tuning;")
    body = src[src.rindex("typedef enum {", 0, end) + len("typedef enum {"):end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n.strip().split("=")[0].strip() for n in body.split(",") if n.strip()]
    assert names[-1] == "SIFT3D_TUNE_COUNT" and len(names) >= 13
    for value, name in enumerate(names[:-1]):
        assert getattr(built, name.replace("SIFT3D_", "")) == value, name
    extra = [n for n in dir(built) if n.startswith("TUNE_") and "SIFT3D_" + n not in names]
    assert not extra, extra


def test_product_reads_no_environment_variable(built):
    """No switch of the library hides in the caller's environment: the HIP translation units do not call getenv (the
    knobs are sift3d_set_tuning), and their objects do not import the symbol (sort_scan.o does: rocPRIM's own headers)."""
    csrc = os.path.join(ROOT, "3d_sift_cuda_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    for obj in ("api_context.o", "api_timing.o", "api_ops.o", "api_pipeline.o", "api_slab.o", "kernels_volume.o", "kernels_blur_fused.o", "kernels_keypoint.o", "gauss_taps.o"):
        nm = subprocess.run(["nm", "--undefined-only", os.path.join(csrc, "_build", obj)], capture_output=True, text=True)
        assert nm.returncode == 0 and "getenv" not in nm.stdout, obj


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
                # the one permitted mention: the CLI's no-device error text tells the user where the CPU route is (round-5 review)
                t = t.replace("oracle/_build/featExtract_oracle in this repository", "")
                if re.search(r"sift3d_oracle|oracle/|import _oracle|o3_[a-z]+\(", t):
                    bad.append(f)
    assert not bad, bad


def test_gauss_taps_host(built, oracle):
    for s in (0.5, 0.95, 1.2262736558914185, 1.5198684930801392, 3.0900158882141113, 0.0):
        a = built.gauss_taps(s)
        b = oracle.taps(s)
        assert len(a) == len(b) and (a.view(np.uint32) == b.view(np.uint32)).all()


def test_gauss_taps_of_the_shipped_binary_build(built):
    """sift3d_set_libm_variant(SIFT3D_LIBM_GCC5): exp() of a float evaluated as the toolchain of the reference's shipped CPU binary
    did (the C exp(double), the tap's product in double) -- the bits of the oracle's -DO3_REFBIN_VARIANT build, which reproduces
    that binary's .key files byte for byte (tests/test_oracle_pins.py::test_refbin_variant_is_byte_identical); the default stays
    the current-g++ reading, and an unknown value is refused."""
    import _oracle
    orb, o = _oracle.load_refbin(), _oracle.load()
    assert built.set_libm_variant(built.LIBM_GCC5) == built.LIBM_CURRENT
    try:
        differ = 0
        for s in (0.5, 0.95, 1.2262736558914185, 1.2489995956420898, 1.5198684930801392, 1.5450079441070557, 1.9465880393981934,
                  2.452547311782837, 3.0900158882141113, 4.9, 0.0):
            a, b = built.gauss_taps(s), orb.taps(s)
            assert len(a) == len(b) and (a.view(np.uint32) == b.view(np.uint32)).all(), s
            differ += int((a.view(np.uint32) != o.taps(s).view(np.uint32)).sum())
        assert differ > 0                      # the two builds do differ (1.5199 and 3.09 among these)
        with pytest.raises(built.Sift3DError):
            built.set_libm_variant(7)
    finally:
        assert built.set_libm_variant(built.LIBM_CURRENT) == built.LIBM_GCC5
    s = 1.5198684930801392
    assert (built.gauss_taps(s).view(np.uint32) == o.taps(s).view(np.uint32)).all()


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
    # round 6: the one long option this build adds (--libm=gcc5 | current); anything else behind "--" is an unknown argument
    r = subprocess.run([built.FEATEXTRACT, "--libm=gcc4", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 255 and "Error: unknown command line argument: --libm=gcc4" in r.stdout
    # without a HIP device the command line says so, names the CPU route and exits: no CPU path hides in the product
    if built.device_count() <= 0:
        nii = str(tmp_path / "dev.nii")
        built.write_nifti(nii, np.ones((8, 8, 8), np.float32))
        r = subprocess.run([built.FEATEXTRACT, "--libm=gcc5", nii, str(tmp_path / "dev.key")], capture_output=True, text=True)
        assert r.returncode == 255 and "no usable HIP device" in r.stderr and "featExtract_oracle" in r.stderr
        assert not os.path.exists(str(tmp_path / "dev.key"))


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


def test_key_writer_in_parallel_blocks_writes_the_serial_bytes(built, oracle, tmp_path):
    """Round 5: the writer formats blocks of 2 048 records in parallel and writes them at their offsets.  30 000 records (15
    blocks, the last one short) with the eigenvalue filter dropping a third of them, written with 1, 3 and 8 threads
    (OMP_NUM_THREADS is read when the process starts): the same bytes, and the oracle writer's (plain fprintf)."""
    import sys
    rng = np.random.default_rng(11)
    n = 30000
    recs = np.zeros(n, built.FEATURE_DTYPE)
    for f in ("x", "y", "z", "scale"):
        recs[f] = (rng.random(n) * 500).astype(np.float32)
    recs["ori"] = (rng.random((n, 9)) * 2 - 1).astype(np.float32)
    recs["eigs"] = (rng.random((n, 3)) * np.array([1.0, 0.3, 0.05])).astype(np.float32)   # a third fails (e1+e2+e3)^3 < 140 e1 e2 e3
    recs["info"] = rng.integers(0, 64, n).astype(np.uint32)
    recs["desc"] = np.argsort(rng.random((n, 64)), axis=1).astype(np.float32)
    # numbers of every length, also the ones libc spells (huge, infinite, NaN), in the first and in the very last record
    odd = np.array([-0.0, 1e-7, 0.9999995, 3.4e38, -3.4e38, 4.1e9, np.inf, -np.inf, np.nan, 123456.789, -2.0 ** 31], np.float32)
    for f in ("x", "y", "z", "scale"):
        at = rng.integers(0, n, 600)
        recs[f][at] = odd[rng.integers(0, len(odd), 600)]
        recs[f][0], recs[f][n - 1] = np.float32(-3.4e38), np.float32(np.inf)
    recs["eigs"][0] = recs["eigs"][n - 1] = (1.0, 1.0, 1.0)            # both kept
    recs["desc"][n - 1] = np.arange(64, dtype=np.float32) - 100.0      # negative (char) values
    ref = str(tmp_path / "ref.key")
    oracle.write_key(ref, recs, comments=["a", "b", "c"])
    want = open(ref, "rb").read()
    kept = int(want.split(b"Features: ")[1].split(b"\n")[0])
    assert 0.2 * n < kept < 0.95 * n and want.count(b"\n") == kept + 6
    np.save(str(tmp_path / "recs.npy"), recs)
    code = ("import importlib, sys, numpy as np; sys.path.insert(0, %r); p = importlib.import_module('3d_sift_cuda_amd'); "
            "p.host_lib().sift3d_write_key_mode(int(sys.argv[3])); "
            "p.write_key(sys.argv[2], np.load(sys.argv[1]), comments=['a', 'b', 'c'])" % ROOT)
    for threads, mode in ((1, 0), (3, 0), (8, 0), (8, 1), (3, 1)):   # mode 0: positional writes; 1: into the mapped file
        out = str(tmp_path / ("t%d_%d.key" % (threads, mode)))
        with open(out, "wb") as f:
            f.write(b"x" * (len(want) + 12345))   # an older, longer file of that name: the writer must leave none of it
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "recs.npy"), out, str(mode)], capture_output=True, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
        assert r.returncode == 0, r.stderr
        assert open(out, "rb").read() == want, (threads, mode)
    # nothing to write, and fewer records than a block
    for m in (0, 1, 2047, 2048, 2049):
        a, b = str(tmp_path / "a.key"), str(tmp_path / "b.key")
        oracle.write_key(a, recs[:m], eig_thres=-1.0, comments=[])
        built.write_key(b, recs[:m], eig_thres=-1.0, comments=[])
        assert open(a, "rb").read() == open(b, "rb").read(), m


def test_key_reader_fast_path_returns_the_fscanf_bits(built, tmp_path):
    """Round 5: sift3d_read_key parses a file in the writers' own layout from memory, in parallel, and gives everything else to
    the fscanf loop (mode 1 forces the loop: the reference's reader restated, pinned in test_oracle_pins.py).  Same bits both ways
    for: 20 000 records of random bit patterns (NaN, infinities and 47-character numbers among them: whole-file fall-back),
    ordinary records (fast path), 30 000 hand-written decimals of up to 15 digits that sit on and beside midpoints of two floats
    (the tokens the fast conversion hands to strtof), layouts the fast path must not take (spaces for tabs, two records a line,
    a record over two lines, an exponent), more lines than the header counts; and the same refusals (a short file, a bad field)."""
    host = built.host_lib()
    rng = np.random.default_rng(21)

    def both(path):
        out = []
        for mode in (0, 1):
            host.sift3d_read_key_mode(mode)
            try:
                out.append(built.read_key(path).tobytes())
            except built.Sift3DError as e:
                out.append(str(e).split("(")[-1])
            finally:
                host.sift3d_read_key_mode(0)
        assert out[0] == out[1], path
        return out[0]

    n = 20000
    recs = np.zeros(n, built.FEATURE_DTYPE)
    for f, k in (("x", 1), ("y", 1), ("z", 1), ("scale", 1), ("ori", 9), ("eigs", 3)):
        recs[f] = rng.integers(0, 1 << 32, n * k, dtype=np.uint64).astype(np.uint32).view(np.float32).reshape(recs[f].shape)
    recs["info"] = rng.integers(0, 1 << 31, n).astype(np.uint32)
    recs["desc"] = rng.integers(-128, 128, (n, 64)).astype(np.float32)
    p1 = str(tmp_path / "bits.key")
    built.write_key(p1, recs, eig_thres=-1.0, comments=["a", "b", "c"])
    assert b"nan" in open(p1, "rb").read() and len(both(p1)) == n * built.FEATURE_DTYPE.itemsize
    fin = recs.copy()
    for f in ("x", "y", "z", "scale", "ori", "eigs"):
        fin[f] = (rng.random(fin[f].shape) * 10.0 ** rng.integers(-6, 9, fin[f].shape) * rng.choice([-1.0, 1.0], fin[f].shape)).astype(np.float32)
    p2 = str(tmp_path / "fin.key")
    built.write_key(p2, fin, eig_thres=-1.0, comments=[])
    got = np.frombuffer(both(p2), built.FEATURE_DTYPE)
    assert (got["desc"] == fin["desc"]).all() and (got["info"] == fin["info"]).all()
    # decimals on and beside the midpoints of neighbouring floats, with few enough digits for the fast conversion
    head = "# featExtract 1.1\nFeatures: %d\nScale-space location[x y z scale] orientation[o11 o12 o13 o21 o22 o23 o31 o32 o32] 2nd moment eigenvalues[e1 e2 e3] info flag[i1] descriptor[d1 .. d64]\n"
    tail = "\t".join(["1.000000"] * 12 + ["16"] + ["%d" % (i % 64) for i in range(64)]) + "\t\n"
    lines = []
    for i in range(7500):
        e = int(rng.integers(-20, 21))
        f0 = np.float32(rng.integers(1 << 23, 1 << 24) * 2.0 ** e)
        mid = (float(f0) + float(np.nextafter(f0, np.float32(np.inf)))) / 2.0     # exact in double
        toks = []
        for digits in (15, 12, 9, 7):
            t = np.format_float_positional(mid, precision=digits, unique=False, fractional=False, trim="-")
            toks.append(t if "e" not in t and len(t.replace(".", "").lstrip("0")) <= 15 else "0.5")
        lines.append("\t".join(toks) + "\t" + tail)
    p3 = str(tmp_path / "mid.key")
    open(p3, "w").write(head % len(lines) + "".join(lines))
    mids = np.frombuffer(both(p3), built.FEATURE_DTYPE)
    assert len(mids) == 7500 and (mids["info"] == 16).all()
    # layouts the fast path must leave to fscanf, and refusals
    text = open(p2).read().split("\n")
    body = text[3:203]
    variants = {"spaces.key": "\n".join(text[:1] + ["Features: 200", text[2]] + [l.replace("\t", " ") for l in body]) + "\n",
                "twoperline.key": "\n".join(text[:1] + ["Features: 200", text[2]] + [body[i] + body[i + 1] for i in range(0, 200, 2)]) + "\n",
                "split.key": "\n".join(text[:1] + ["Features: 200", text[2]] + [l[:40] + "\n" + l[40:] if l[39] == "\t" else l for l in body]) + "\n",
                "exponent.key": "\n".join(text[:1] + ["Features: 200", text[2]] + ["1e2\t" + l.split("\t", 1)[1] for l in body]) + "\n",
                "more.key": "\n".join(text[:1] + ["Features: 150", text[2]] + body) + "\n",
                "short.key": "\n".join(text[:1] + ["Features: 300", text[2]] + body) + "\n",
                "badfield.key": "\n".join(text[:1] + ["Features: 200", text[2]] + body[:100] + ["x" + body[100]] + body[101:]) + "\n"}
    res = {}
    for name, content in variants.items():
        q = str(tmp_path / name)
        open(q, "w").write(content)
        res[name] = both(q)
    want200 = np.frombuffer(both(p2), built.FEATURE_DTYPE)[:200].tobytes()
    for name in ("spaces.key", "twoperline.key", "split.key"):
        assert res[name] == want200, name
    assert res["more.key"] == want200[:150 * built.FEATURE_DTYPE.itemsize]
    assert np.frombuffer(res["exponent.key"], built.FEATURE_DTYPE)["x"].tolist() == [100.0] * 200
    assert res["short.key"] == "-2)" and res["badfield.key"] == "-2)"


def test_synth_slices_are_the_planes_of_the_whole_volume(built):
    """sift3d_synth_blobs_slices (bench.py's Z-slab ranks generate only their input slices): the same bits as those planes of
    the whole volume, for ranges at the faces, inside, empty and clipped; above and below the size where the generator
    goes parallel."""
    for dims in ((40, 36, 50), (200, 180, 170)):
        v = built.synth_blobs(*dims, seed=9)
        nz = dims[2]
        for a, b in ((0, nz), (13, 37), (nz - 9, nz), (0, 1), (20, 20), (-5, 8), (nz - 3, nz + 10)):
            s = built.synth_blobs_slices(dims[0], dims[1], nz, a, b, seed=9)
            lo, hi = max(0, a), min(nz, b)
            assert s.shape == (max(0, hi - lo), dims[1], dims[0]) and s.tobytes() == v[lo:hi].tobytes(), (dims, a, b)


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


# ---------------------------------------------------------------------------------------------------
# nifti_min_read: every source datatype of reg_changeDatatype (R/featExtract/featExtract.cpp:18-77), both byte orders,
# .nii / .nii.gz / .hdr + .img / .hdr + .img.gz.  Files are written here with numpy + struct (not by the product's
# own float32 writer), so the reader is checked against an independent encoder.
# ---------------------------------------------------------------------------------------------------
NIFTI_TYPES = {2: np.uint8, 256: np.int8, 4: np.int16, 512: np.uint16, 8: np.int32, 768: np.uint32, 16: np.float32, 64: np.float64}


def _nifti_bytes(arr, code, big_endian=False, pair=False, voxel=(1.0, 1.5, 2.0), slope=3.0, vox_offset=None, ndim=3, magic=None):
    """(header bytes, voxel bytes) of a NIfTI-1 file holding arr (shape nz, ny, nx) as datatype `code`."""
    import struct
    e = ">" if big_endian else "<"
    nz, ny, nx = arr.shape
    h = bytearray(348)
    struct.pack_into(e + "i", h, 0, 348)
    struct.pack_into(e + "8h", h, 40, ndim, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into(e + "h", h, 70, code)
    struct.pack_into(e + "h", h, 72, 8 * np.dtype(NIFTI_TYPES.get(code, np.uint8)).itemsize)
    struct.pack_into(e + "8f", h, 76, 1.0, voxel[0], voxel[1], voxel[2], 1.0, 1.0, 1.0, 1.0)
    off = 0.0 if pair else 352.0
    struct.pack_into(e + "f", h, 108, off if vox_offset is None else vox_offset)
    struct.pack_into(e + "2f", h, 112, slope, 7.0)          # scl_slope / scl_inter: the reference ignores them
    h[344:348] = (b"ni1\0" if pair else b"n+1\0") if magic is None else magic
    data = np.ascontiguousarray(arr.astype(NIFTI_TYPES.get(code, np.uint8)))
    if big_endian:
        data = data.byteswap()
    return bytes(h), data.tobytes()


def _write_case(tmp_path, name, arr, code, **kw):
    import gzip
    pair = name.endswith(".hdr")
    gz = name.endswith(".gz") or kw.pop("img_gz", False)
    h, d = _nifti_bytes(arr, code, pair=pair, **kw)
    p = str(tmp_path / name)
    if pair:
        open(p, "wb").write(h)
        ip = p[:-4] + ".img"
        if gz:
            gzip.open(ip + ".gz", "wb").write(d)
        else:
            open(ip, "wb").write(d)
    else:
        blob = h + b"\0\0\0\0" + d
        (gzip.open(p, "wb") if gz else open(p, "wb")).write(blob)
    return p


def _test_values(code, shape, rng):
    dt = np.dtype(NIFTI_TYPES[code])
    if dt.kind == "f":
        v = (rng.standard_normal(shape) * 10.0 ** rng.integers(-3, 6, shape)).astype(dt)
    else:
        info = np.iinfo(dt)
        v = rng.integers(info.min, int(info.max) + 1, shape, dtype=np.int64).astype(dt)
        v.flat[:4] = [info.min, info.max, 0, info.max // 2 + 1]      # extremes: sign handling, values above 2^24 round
    return v


@pytest.mark.parametrize("code", sorted(NIFTI_TYPES))
def test_nifti_reader_all_datatypes(built, tmp_path, code):
    rng = np.random.default_rng(code)
    shape = (5, 7, 9)
    v = _test_values(code, shape, rng)
    want = v.astype(np.float32)                                      # the plain C cast of reg_changeDatatype
    for name, kw in (("le.nii", {}), ("be.nii", {"big_endian": True}), ("le.nii.gz", {}), ("be.nii.gz", {"big_endian": True}),
                     ("p_le.hdr", {}), ("p_be.hdr", {"big_endian": True}), ("p_gz.hdr", {"img_gz": True})):
        p = _write_case(tmp_path, name, v, code, **kw)
        got, hdr = built.read_nifti(p)
        assert hdr["dims"] == (9, 7, 5, 1) and hdr["datatype"] == code, name
        assert hdr["voxel"] == (1.0, 1.5, 2.0), name
        assert got.shape == shape and got.dtype == np.float32
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), (name, code)   # scl_slope / scl_inter ignored
        assert hdr["qform_code"] == 0 and hdr["sform_code"] == 0
        assert np.allclose(np.diag(hdr["qto_xyz"]), [1.0, 1.5, 2.0, 1.0])            # method 1: voxel scaling only


@pytest.mark.parametrize("code", [4, 512, 64, 2])
def test_nifti_reader_casts_a_large_run_in_parallel(built, tmp_path, code):
    """From 2^20 voxels on the cast to float runs over the host cores (csrc/nifti_min.c): the same floats as numpy's cast."""
    rng = np.random.default_rng(code)
    v = _test_values(code, (70, 128, 128), rng)
    for name, kw in (("big.nii", {}), ("big_be.nii.gz", {"big_endian": True})):
        got, _ = built.read_nifti(_write_case(tmp_path, name, v, code, **kw))
        assert (got.view(np.uint32) == v.astype(np.float32).view(np.uint32)).all(), name


def test_nifti_gz_in_one_call_and_through_zlib_give_the_same_voxels(built, tmp_path):
    """A gzip'ed data file is inflated in one call by libdeflate where the system has it (csrc/nifti_min.c: fast_inflate; 2 - 3 x zlib's
    streaming inflate, which is what `featExtract in.nii.gz` spends its time in) and by zlib otherwise -- and whenever the file is not
    what that path expects: several gzip members, bytes behind the stream, a stream cut short.  Every datatype, both byte orders,
    single files and .hdr + .img.gz pairs: the same voxels, the same return codes."""
    import ctypes.util
    import gzip
    have = ctypes.util.find_library("deflate") is not None
    rng = np.random.default_rng(5)
    cases = []
    for code in sorted(NIFTI_TYPES):
        v = _test_values(code, (6, 10, 12), rng)
        cases.append((_write_case(tmp_path, "a%d.nii.gz" % code, v, code), v))
        cases.append((_write_case(tmp_path, "b%d.nii.gz" % code, v, code, big_endian=True), v))
        cases.append((_write_case(tmp_path, "c%d.hdr" % code, v, code, img_gz=True), v))
    v = _test_values(4, (9, 8, 7), rng)
    cases.append((_write_case(tmp_path, "off.nii.gz", v, 4, vox_offset=400.0, slope=1.0), None))    # a header that promises 48 bytes more than the file has
    h, d = _nifti_bytes(v, 4)
    blob = h + b"\0\0\0\0" + d
    two = str(tmp_path / "two_members.nii.gz")                                                  # one file, two gzip members
    open(two, "wb").write(gzip.compress(blob[:1000]) + gzip.compress(blob[1000:]))
    tail = str(tmp_path / "tail.nii.gz")                                                        # bytes behind the stream
    open(tail, "wb").write(gzip.compress(blob) + b"not gzip")
    cut = str(tmp_path / "cut.nii.gz")                                                          # a stream cut short
    open(cut, "wb").write(gzip.compress(blob)[:-40])
    longer = str(tmp_path / "longer.nii.gz")                                                    # more bytes than the header's voxels
    open(longer, "wb").write(gzip.compress(blob + b"\x07" * 1000))

    def read(p):
        try:
            return built.read_nifti(p)[0]
        except built.Sift3DError as e:
            return str(e)
    results = {}
    for on in (1, 0):
        built.nifti_fast_inflate(on)
        n0 = built.nifti_fast_inflate_count()
        results[on] = [read(p) for p, _ in cases] + [read(two), read(tail), read(cut), read(longer)]
        taken = built.nifti_fast_inflate_count() - n0
        if on and have:
            assert taken == len(cases), taken              # every well-formed single-member file and `longer`; never off / two / tail / cut
        else:
            assert taken == 0
    built.nifti_fast_inflate(1)
    for a, b, (p, want) in zip(results[1], results[0], cases + [(two, None), (tail, None), (cut, None), (longer, None)]):
        assert type(a) is type(b), p
        if isinstance(a, str):
            assert a == b, p
        else:
            assert a.tobytes() == b.tobytes(), p
            if want is not None:
                assert (a.view(np.uint32) == want.astype(np.float32).view(np.uint32)).all(), p
    assert isinstance(results[1][-2], str) and not isinstance(results[1][-1], str) and not isinstance(results[1][-4], str)


def test_nifti_reader_rejects_what_it_cannot_read(built, tmp_path):
    v = np.arange(4 * 5 * 6, dtype=np.int16).reshape(4, 5, 6)

    def code_of(name, **kw):
        p = _write_case(tmp_path, name, v, kw.pop("code", 4), **kw)
        try:
            built.read_nifti(p)
        except built.Sift3DError as e:
            return e.code
        return 0
    assert code_of("ok.nii") == 0
    assert code_of("rgb.nii", code=128) == -3                       # DT_RGB24: not a scalar type
    assert code_of("cplx.nii", code=32) == -3
    assert code_of("dim0.nii", ndim=0) == -1 and code_of("dim9.nii", ndim=9) == -1
    p = _write_case(tmp_path, "short.nii", v, 4)
    open(p, "r+b").truncate(352 + 100)                               # voxel data cut short
    with pytest.raises(built.Sift3DError) as ei:
        built.read_nifti(p)
    assert ei.value.code == -2
    open(str(tmp_path / "tiny.nii"), "wb").write(b"\x5c\x01\0\0" + b"\0" * 40)   # header cut short
    with pytest.raises(built.Sift3DError) as ei:
        built.read_nifti(str(tmp_path / "tiny.nii"))
    assert ei.value.code == -1
    h, d = _nifti_bytes(v, 4)
    open(str(tmp_path / "badsize.nii"), "wb").write(b"\x01\x02\x03\x04" + h[4:] + b"\0\0\0\0" + d)   # sizeof_hdr is not 348 in either byte order
    with pytest.raises(built.Sift3DError) as ei:
        built.read_nifti(str(tmp_path / "badsize.nii"))
    assert ei.value.code == -1
    import struct
    h = bytearray(h); struct.pack_into("<h", h, 42, -3)              # a negative row length
    open(str(tmp_path / "negx.nii"), "wb").write(bytes(h) + b"\0\0\0\0" + d)
    with pytest.raises(built.Sift3DError) as ei:
        built.read_nifti(str(tmp_path / "negx.nii"))
    assert ei.value.code == -1
    open(str(tmp_path / "lonely.hdr"), "wb").write(_nifti_bytes(v, 4, pair=True)[0])   # header without its .img
    with pytest.raises(built.Sift3DError) as ei:
        built.read_nifti(str(tmp_path / "lonely.hdr"))
    assert ei.value.code == -2
    with pytest.raises(built.Sift3DError):
        built.read_nifti(str(tmp_path / "does_not_exist.nii"))


def test_nifti_reader_vox_offset_and_analyze(built, tmp_path):
    """A single file whose voxels start beyond byte 352 (header extensions), and an Analyze-7.5 header (no NIfTI magic:
    qform / sform codes are not trusted)."""
    v = np.arange(3 * 4 * 5, dtype=np.float32).reshape(3, 4, 5) - 17.5
    h, d = _nifti_bytes(v, 16, vox_offset=352.0 + 64)
    open(str(tmp_path / "ext.nii"), "wb").write(h + b"\x01\0\0\0" + b"\xAA" * 64 + d)
    got, _ = built.read_nifti(str(tmp_path / "ext.nii"))
    assert (got == v).all()
    import struct
    h = bytearray(_nifti_bytes(v, 16, pair=True, magic=b"\0\0\0\0")[0])
    struct.pack_into("<2h", h, 252, 2, 3)                            # garbage where NIfTI keeps the form codes
    open(str(tmp_path / "an.hdr"), "wb").write(bytes(h))
    open(str(tmp_path / "an.img"), "wb").write(d)
    got, hdr = built.read_nifti(str(tmp_path / "an.hdr"))
    assert (got == v).all() and hdr["qform_code"] == 0 and hdr["sform_code"] == 0


def test_slab_driver_host_threads_under_thread_sanitizer():
    """The one-process slab driver runs every rank's launches on a host thread of its own (csrc/zs_crew.h: a step is published to
    all workers, acknowledged by all, and what a step reads of a neighbour was written a step earlier).  tests/crew_check.cpp steps
    such a crew through tens of thousands of steps on plain memory under ThreadSanitizer (`make tsan`), with workers spinning, asleep
    and outnumbering the cores."""
    csrc = os.path.join(ROOT, "3d_sift_cuda_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    exe = os.path.join(csrc, "_build", "crew_check_tsan")
    for workers, steps in ((1, 5000), (3, 20000), (7, 10000), (15, 3000)):
        r = subprocess.run([exe, str(workers), str(steps)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
        assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr and "crew ok: %d workers, %d steps" % (workers, steps) in r.stdout, \
            r.stdout[-500:] + r.stderr[-3000:]


def test_host_code_under_sanitizers(built, tmp_path):
    """ASan + UBSan over the plain-C host code (nifti_min.c, world.c, keyfile.c, synth.c, match_votes.c: `make asan`) on every file shape
    above plus malformed ones, and over the oracle CLI (restatement + reader) on a small volume with the -2+ / -b / -w
    options.  A sanitizer report makes the process exit non-zero."""
    csrc = os.path.join(ROOT, "3d_sift_cuda_amd", "csrc")
    for d in (csrc, os.path.join(ROOT, "oracle")):
        r = subprocess.run(["make", "-C", d, "asan"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    drv = os.path.join(csrc, "_build", "host_asan_driver")
    rng = np.random.default_rng(0)
    files = []
    for code in sorted(NIFTI_TYPES):
        v = _test_values(code, (6, 5, 8), rng)
        files.append(_write_case(tmp_path, "t%d.nii" % code, v, code, voxel=(1.0, 1.25, 2.0)))
        files.append(_write_case(tmp_path, "b%d.nii.gz" % code, v, code, big_endian=True))
        files.append(_write_case(tmp_path, "p%d.hdr" % code, v, code, img_gz=(code % 3 == 0)))
    v = np.zeros((4, 4, 4), np.int16)
    files.append(_write_case(tmp_path, "rgb.nii", v, 128))
    p = _write_case(tmp_path, "cut.nii", v, 4); open(p, "r+b").truncate(400); files.append(p)
    open(str(tmp_path / "junk.nii"), "wb").write(bytes(rng.integers(0, 256, 500, dtype=np.uint8))); files.append(str(tmp_path / "junk.nii"))
    h, d = _nifti_bytes(v, 4, voxel=(0.0, -1.0, float("nan")))       # degenerate voxel sizes through the resampler
    open(str(tmp_path / "vox.nii"), "wb").write(h + b"\0\0\0\0" + d); files.append(str(tmp_path / "vox.nii"))
    files.append(str(tmp_path / "absent.nii"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([drv, "read"] + files, capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    read_rc = {os.path.basename(l.split(" rc=")[0]): int(l.split(" rc=")[1].split()[0]) for l in r.stdout.splitlines()}
    assert len(read_rc) == len(files) and sum(1 for v in read_rc.values() if v == 0) == 25   # the 24 well-formed files and vox.nii
    assert read_rc["rgb.nii"] == -3 and read_rc["cut.nii"] == -2 and read_rc["junk.nii"] == -1 and read_rc["absent.nii"] == -1
    assert "vox.nii rc=0" in r.stdout and "iso rc=-3" in r.stdout    # unusable voxel sizes are refused by the resampler, not crashed on
    r = subprocess.run([drv, "keys", str(tmp_path)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "keys ok" in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    r = subprocess.run([drv, "votes"], capture_output=True, text=True, env=env)   # round 4: the matcher's host side, incl. refused inputs
    assert r.returncode == 0 and "votes ok" in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stdout + r.stderr[-3000:]
    cli = os.path.join(ROOT, "oracle", "_build", "featExtract_oracle_asan")
    nii = str(tmp_path / "s.nii")
    built.write_nifti(nii, built.synth_blobs(40, 36, 32, seed=5), voxel=(1.0, 1.25, 1.5), qform=(0.1, 0.2, 0.3, -3.0, 2.0, 5.0, -1.0))
    for flags in ([], ["-2+", "-b"], ["-2-", "-bn"], ["-w", "-br"]):
        r = subprocess.run([cli] + flags + [nii, str(tmp_path / "s.key")], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (flags, r.stderr[-3000:])
