"""The matcher: exact nearest neighbours on the GPU (sift3d_knn64), the vote accumulation on the
host, the featMatchMultiple command line -- against the CPU restatement in oracle/match_oracle.c.  Integer distances and
indices must agree exactly (order: ascending distance, ties to the lower index); votes are float sums taken in the same
order on both sides, so they agree bit for bit too."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_descriptors(rng, n):
    """n random rank descriptors: permutations of 0..63, as NormalizeDataRankedPCs leaves them."""
    return np.argsort(rng.random((n, 64)), axis=1).astype(np.int8)


def clustered(rng, n, n_centres, swaps):
    """Rank descriptors around a few centres (a handful of transpositions away): many near neighbours, many exact ties."""
    centres = rank_descriptors(rng, n_centres)
    out = centres[rng.integers(0, n_centres, n)].copy()
    for row in out:
        for _ in range(int(rng.integers(0, swaps + 1))):
            a, b = rng.integers(0, 64, 2)
            row[a], row[b] = row[b], row[a]
    return out


# ---- host side, no GPU -----------------------------------------------------------------------------------------------
def test_votes_match_the_oracle_on_cpu(built, oracle):
    rng = np.random.default_rng(3)
    sizes = [40, 0, 57, 23, 64]
    first = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    desc = clustered(rng, int(first[-1]), 9, 3)
    k = 6
    idx, d2 = oracle.knn64(desc, desc, k)
    labels = np.arange(len(sizes), dtype=np.int32)
    v0, c0 = oracle.match_votes(first, labels, len(sizes), idx, d2)
    v1, c1 = built.match_votes(first, labels, len(sizes), idx, d2)
    assert (c0 == c1).all() and c0.sum() > 20
    assert v0.tobytes() == v1.tobytes() and (v0 > 0).any()
    assert (np.diag(c0) == 0).all()                       # an image never votes for itself
    # fewer labels than images (two images share a label)
    labels2 = np.array([0, 1, 0, 2, 1], np.int32)
    v0, c0 = oracle.match_votes(first, labels2, 3, idx, d2)
    v1, c1 = built.match_votes(first, labels2, 3, idx, d2)
    assert (c0 == c1).all() and v0.tobytes() == v1.tobytes()


def test_votes_refuse_indices_out_of_range_and_are_the_same_for_any_thread_count(built, oracle):
    """advisor, round 3: sift3d_match_votes is public C-ABI -- a neighbour index beyond the features or a label beyond
    n_labels must be an error, not a read or write out of bounds.  Round 4: the loop over query images runs under OpenMP
    as the reference's does (featMatchMultiple.cpp:108); rows must not depend on the thread count."""
    rng = np.random.default_rng(8)
    sizes = [30] * 12
    first = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    desc = clustered(rng, int(first[-1]), 7, 2)
    idx, d2 = oracle.knn64(desc, desc, 5)
    labels = (np.arange(len(sizes)) % 4).astype(np.int32)
    v0, c0 = oracle.match_votes(first, labels, 4, idx, d2)
    v1, c1 = built.match_votes(first, labels, 4, idx, d2)
    assert v0.tobytes() == v1.tobytes() and (c0 == c1).all() and c0.sum() > 50
    code = ("import importlib, numpy as np, sys; p = importlib.import_module('3d_sift_cuda_amd'); d = np.load(sys.argv[1]); "
            "v, c = p.match_votes(d['first'], d['labels'], 4, d['idx'], d['d2']); sys.stdout.buffer.write(v.tobytes() + c.tobytes())")
    import subprocess, sys, os, tempfile
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "a.npz"), first=first, labels=labels, idx=idx, d2=d2)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for nt in ("1", "3", "8"):
            r = subprocess.run([sys.executable, "-c", code, os.path.join(td, "a.npz")], capture_output=True, cwd=root,
                               env=dict(os.environ, OMP_NUM_THREADS=nt))
            assert r.returncode == 0, r.stderr
            assert r.stdout == v0.tobytes() + c0.tobytes()
    bad = idx.copy()
    bad[7, 2] = int(first[-1])            # one past the last feature
    with pytest.raises(built.Sift3DError):
        built.match_votes(first, labels, 4, bad, d2)
    with pytest.raises(built.Sift3DError):
        built.match_votes(first, labels, 3, idx, d2)          # label 3 with n_labels 3
    neg = labels.copy()
    neg[2] = -1
    with pytest.raises(built.Sift3DError):
        built.match_votes(first, neg, 4, idx, d2)


def test_filters_and_descriptor_bytes(built):
    f = np.zeros(6, built.FEATURE_DTYPE)
    f["info"] = [0x00, 0x10, 0x20, 0x30, 0x20, 0x00]
    f["x"] = np.arange(6)
    f["desc"] = np.arange(64, dtype=np.float32)[None, :]
    f["ori"][:, 1] = 0.5
    keep = built.match_filter(f, reoriented=1, peaks=4)
    assert list(keep["x"]) == [2, 3, 4] and (keep["ori"][:, 1] == 0.5).all()
    assert list(built.match_filter(f, reoriented=1, peaks=0)["x"]) == [2, 4]      # peaks: MIN0MAX1 clear
    assert list(built.match_filter(f, reoriented=1, peaks=1)["x"]) == [3]         # valleys
    plain = built.match_filter(f, reoriented=0, peaks=4)
    assert list(plain["x"]) == [0, 1, 5] and (plain["ori"] == np.eye(3, dtype=np.float32).ravel()).all()
    d = built.match_descriptors(f)
    assert d.dtype == np.int8 and (d == np.arange(64)).all()
    f["desc"][0, 3] = 200.0
    with pytest.raises(built.Sift3DError):
        built.match_descriptors(f)


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n_db,n_q,k", [(1000, 1000, 5), (129, 300, 8), (5, 40, 8), (4097, 257, 16), (700, 128, 1), (2500, 1, 32), (300, 33, 20)])
def test_knn_is_exact(built, oracle, n_db, n_q, k):
    """Sizes that are not multiples of the 128-vector tiles, fewer database vectors than k, one query, every list length."""
    rng = np.random.default_rng(n_db * 7 + k)
    db = clustered(rng, n_db, 11, 4)
    q = np.concatenate([db[: min(n_q, n_db) // 2], clustered(rng, n_q - min(n_q, n_db) // 2, 11, 4)])   # half of the queries are in the database
    want_i, want_d = oracle.knn64(db, q, k)
    got_i, got_d, ms = built.knn64(db, q, k)
    assert (got_d == want_d).all()
    assert (got_i == want_i).all()                                        # ties resolved to the lower index on both sides
    assert (got_d[: min(n_q, n_db) // 2, 0] == 0).all()                   # a vector of the database finds itself


@pytest.mark.gpu
@pytest.mark.parametrize("k", [3, 8, 20])
def test_knn_when_every_row_is_a_candidate(built, oracle, k):
    """The order a search likes least: every database row is nearer to the queries than all rows before it, so every row beats
    the running k-th distance (the lanes' queues are full all the time and candidates go straight into the lists); the
    second half of the database repeats rows of the first (ties between the two lanes of a pair, resolved by index)."""
    rng = np.random.default_rng(77 + k)
    base = np.argsort(rng.random(64)).astype(np.int8)
    rows = []
    for i in range(1500):
        v = base.copy()
        for _ in range((1500 - i) // 30):                                  # fewer transpositions as the file goes on
            a, b = rng.integers(0, 64, 2)
            v[a], v[b] = v[b], v[a]
        rows.append(v)
    db = np.stack(rows + [rows[i] for i in rng.integers(0, 1500, 1300)])   # 2 800 rows: not a multiple of the tile
    q = np.stack([base] + [db[i] for i in rng.integers(0, 2800, 199)])
    want_i, want_d = oracle.knn64(db, q, k)
    got_i, got_d, _ = built.knn64(db, q, k)
    assert (got_d == want_d).all() and (got_i == want_i).all()


@pytest.mark.gpu
def test_knn_few_queries_against_a_long_database(built, oracle):
    """Few queries: the database is cut into segments searched by different workgroups (sift3d_knn_plan) and the merge
    kernel joins 2 x segments lists per query -- same neighbours, same order on ties (every vector is there four times)."""
    rng = np.random.default_rng(4242)
    quarter = clustered(rng, 10_100, 37, 5)
    db = np.concatenate([quarter, quarter, quarter, quarter])             # 40 400 rows = 158 tiles: 4 segments for one block of queries
    q = np.concatenate([quarter[:40], clustered(rng, 30, 37, 5)])
    for k in (5, 12):
        want_i, want_d = oracle.knn64(db, q, k)
        got_i, got_d, _ = built.knn64(db, q, k)
        assert (got_d == want_d).all() and (got_i == want_i).all()
    assert (got_d[:40, :4] == 0).all() and (np.diff(got_i[:40, :4], axis=1) > 0).all()         # the copies, lowest index first
    gen = rng.integers(0, 128, (40_100, 64)).astype(np.int8)              # norms differ: the general kernel, segments again, a ragged last tile
    gq = np.concatenate([gen[100:130], rng.integers(0, 128, (25, 64)).astype(np.int8)])
    want_i, want_d = oracle.knn64(gen, gq, 9)
    got_i, got_d, _ = built.knn64(gen, gq, 9)
    assert (got_d == want_d).all() and (got_i == want_i).all()


@pytest.mark.gpu
def test_knn_extreme_components(built, oracle):
    """The largest distances the byte range allows (64 x 127^2): the lists' keys (distance x 2^31 + index, held in doubles)
    must stay exact, and rows past the end of the database (zero vectors inside the last tile) must stay out."""
    rng = np.random.default_rng(9)
    db = np.zeros((700, 64), np.int8)
    db[::2] = 127                                                          # all-127 and all-zero rows: norms 1 032 256 and 0
    db[5::7, :32] = 0
    db[100:140] = rng.integers(0, 128, (40, 64))
    q = np.concatenate([db[:70], np.zeros((3, 64), np.int8), np.full((3, 64), 127, np.int8), rng.integers(0, 128, (60, 64)).astype(np.int8)])
    for k in (1, 6, 32):
        want_i, want_d = oracle.knn64(db, q, k)
        got_i, got_d, _ = built.knn64(db, q, k)
        assert (got_d == want_d).all() and (got_i == want_i).all()
    same = np.full((260, 64), 9, np.int8)                                  # one norm for all (the constant-norm kernel), all distances 0
    gi, gd, _ = built.knn64(same, same[:5], 8)
    assert (gd == 0).all() and (gi == np.arange(8)[None, :]).all()


@pytest.mark.gpu
def test_knn_general_int8_vectors_and_errors(built, oracle):
    rng = np.random.default_rng(5)
    db = rng.integers(0, 128, (900, 64)).astype(np.int8)                  # not permutations: the norms differ
    q = rng.integers(0, 128, (130, 64)).astype(np.int8)
    wi, wd = oracle.knn64(db, q, 7)
    gi, gd, _ = built.knn64(db, q, 7)
    assert (gi == wi).all() and (gd == wd).all()
    bad = db.copy(); bad[3, 5] = -1
    with pytest.raises(built.Sift3DError):
        built.knn64(bad, q, 3)
    with pytest.raises(built.Sift3DError):
        built.knn64(db, q, 33)


@pytest.mark.gpu
def test_matcher_command_line(built, oracle, tmp_path):
    """featMatchMultiple on .key files written by featExtract's writer: four 'images' that share blobs (the same volume
    with blocks of it replaced), neighbours 5.  matching_votes.txt and vote_count.txt must be what the oracle's search and
    vote accumulation give on the same filtered records."""
    dims = (72, 64, 56)
    base = built.synth_blobs(*dims, seed=77)
    vols = [base.copy() for _ in range(4)]
    other = built.synth_blobs(*dims, seed=78)
    vols[1][:, :, 36:] = other[:, :, 36:]
    vols[2][28:] = other[28:]
    vols[3] = other
    names, sets = [], []
    with built.Context(*dims) as ctx:
        for i, v in enumerate(vols):
            ctx.set_volume(v)
            f = ctx.extract()
            p = str(tmp_path / ("img%d.key" % i))
            built.write_key(p, f)
            names.append(p)
            sets.append(built.match_filter(built.read_key(p), reoriented=1, peaks=4))
    r = subprocess.run([built.FEATMATCH, "-n", "5"] + names, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Creating NN index structure, NN=5, image split=-1, type features=Peak and Valley" in r.stdout
    first = np.concatenate([[0], np.cumsum([len(s) for s in sets])]).astype(np.int64)
    allf = np.concatenate(sets)
    desc = built.match_descriptors(allf)
    idx, d2 = oracle.knn64(desc, desc, 5)
    votes, counts = oracle.match_votes(first, np.arange(4, dtype=np.int32), 4, idx, d2)
    lines = open(tmp_path / "matching_votes.txt").read().split("\n")
    assert lines[0] == "Peak and Valley"
    got_v = np.array([[float(x) for x in l.split("\t") if x] for l in lines[1:5]], np.float64)
    want_v = np.array([[float("%f" % x) for x in row] for row in votes])
    assert (got_v == want_v).all() and got_v.sum() > 0
    cl = open(tmp_path / "vote_count.txt").read().split("\n")
    got_c = np.array([[int(x) for x in l.split("\t") if x] for l in cl[1:5]])
    assert (got_c == counts).all()
    assert got_c[0, 1] > got_c[0, 3] and got_c[0, 2] > got_c[0, 3]         # shared blobs attract votes, a foreign volume few
    fc = open(tmp_path / "feature_count.txt").read().split()
    assert [int(x) for x in fc[1::2]] == [len(s) for s in sets]
    # advisor, round 3: a key file that cannot be read keeps its index and stays an empty set (the reference's loop,
    # featMatchMultiple.cpp:578-632) -- also when it is the LAST name: one row per name in every output
    r = subprocess.run([built.FEATMATCH, "-n", "5"] + names + [str(tmp_path / "missing.key")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "Error: could not open feature file 4" in r.stdout
    fc = open(tmp_path / "feature_count.txt").read().split()
    assert [int(x) for x in fc[1::2]] == [len(s) for s in sets] + [0]
    lines = open(tmp_path / "matching_votes.txt").read().split("\n")
    rows5 = [[float(x) for x in l.split("\t") if x] for l in lines[1:6]]
    assert all(len(row) == 5 for row in rows5) and rows5[4] == [0.0] * 5
    assert (np.array(rows5)[:4, :4] == want_v).all()
    assert len(open(tmp_path / "_names.txt").read().splitlines()) == 5
    # -s 2: all three passes APPEND to the vote files (featMatchMultiple.cpp:58-65), the first included
    before = open(tmp_path / "matching_votes.txt").read()
    r = subprocess.run([built.FEATMATCH, "-n", "5", "-s2"] + names, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    after = open(tmp_path / "matching_votes.txt").read()
    assert after.startswith(before) and [l for l in after[len(before):].split("\n") if l and not l[0].isdigit()] == ["Peak and Valley", "Peaks", "Valley"]
    # usage and bad option, as the reference
    assert subprocess.run([built.FEATMATCH], capture_output=True).returncode == 255
    bad = subprocess.run([built.FEATMATCH, "-q", names[0], names[1]], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode == 255 and "Error: unknown command line argument: -q" in bad.stdout
