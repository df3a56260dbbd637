#!/usr/bin/env python3
"""GPU box: the matcher's search as its own workload.  All-to-all exact k-nearest neighbours over
the rank descriptors of `images` x `per_image` features (defaults: 100 x 2 000 = 200 000 descriptors of 64 int8
components, k = 5: a featMatchMultiple run over a hundred 512^3-sized .key files keeps about that many re-oriented records).
Prints ONE JSON line: value = descriptor pairs compared per second; roofline against the dense int8 MFMA peak
(2 x 64 integer operations per pair); cpu_baseline = the brute-force restatement (oracle/match_oracle.c, OpenMP build)
on a bounded sample of the same queries.
usage: python tools/bench_match.py [images=100] [per_image=2000] [k=5] [repeats=5] [random=0]
KNN_PLAN=groups,segments in the environment: load the development build (make -C 3d_sift_cuda_amd/csrc DEV=1) and override how
the search is cut (query groups per wavefront, database segments); random=1: uniformly random permutations instead of
perturbed copies (the worst case for the per-lane thresholds)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("3d_sift_cuda_amd")
import _oracle
plan = os.environ.get("KNN_PLAN")
ahead = os.environ.get("KNN_AHEAD")   # 2: the development build's two-subtiles-ahead form of the search kernel
if ahead and not plan:
    plan = "0,0"
if plan:
    import ctypes
    pkg.LIB_HIP = os.path.join(pkg.CSRC, os.environ.get("KNN_BUILD", "_build_dev"), "libsift3d_hip.so")   # KNN_BUILD: another development build
    pkg.hip_lib().sift3d_dev_knn_plan(*[ctypes.c_int(int(v)) for v in plan.split(",")])
    if ahead:
        pkg.hip_lib().sift3d_dev_knn_ahead(ctypes.c_int(int(ahead)))
images = int(sys.argv[1]) if len(sys.argv) > 1 else 100
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
uniform = len(sys.argv) > 5 and int(sys.argv[5]) != 0
n = images * per
rng = np.random.default_rng(2026)
centres = np.argsort(rng.random((per, 64)), axis=1).astype(np.int8)          # every image sees (a perturbed copy of) the same anatomy
desc = np.repeat(centres[None], images, axis=0).reshape(n, 64).copy()
if uniform: desc = np.argsort(rng.random((n, 64)), axis=1).astype(np.int8)
for i in ([] if uniform else np.nonzero(rng.random(n) < 0.9)[0]):               # most copies differ by a few transpositions
    row = desc[i]                                                               # (a view: the first version of this loop swapped inside a copy)
    for _ in range(int(rng.integers(1, 6))):
        a, b = rng.integers(0, 64, 2)
        row[a], row[b] = row[b], row[a]
MFMA_I8_PEAK_TOPS = 5000.0   # dense int8, 2 x the bf16 rate (MI355X_MICROARCH.md, Matrix cores)
idx, d2, ms = pkg.knn64(desc, desc, k, repeats=reps + 1)
pairs = float(n) * n
ops = 2.0 * 64 * pairs
m = min(n, 2000)
orc = _oracle.load_omp()
threads = min(len(os.sched_getaffinity(0)), 16)
os.environ.setdefault("OMP_NUM_THREADS", str(threads))
t0 = time.perf_counter(); wi, wd = orc.knn64(desc, desc[:m], k); cdt = time.perf_counter() - t0
same = bool((wi == idx[:m]).all() and (wd == d2[:m]).all())
print(json.dumps({
    "metric": "descriptor pairs compared per second (exact all-to-all %d-NN over 64-component int8 rank descriptors)" % k,
    "value": round(pairs / (ms * 1e-3), 1), "unit": "pairs/s", "n_gpus": 1, "ms_per_step": round(ms, 3), "higher_is_better": True,
    "dtype": "int8 (int32 accumulation)", "data": "synthetic",
    "config": {"workload": "%d images x %d descriptors = %d database vectors = queries, k = %d" % (images, per, n, k)},
    "roofline": {"bound": "mfma", "achieved": round(ops / (ms * 1e-3) / 1e12, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                 "frac": round(ops / (ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, 4), "traffic": None,
                 "kernel": "knn_search_kernel<list length, constant norm> (v_mfma_i32_32x32x32_i8 Gram tiles + per-lane top-k) + norms + merge",
                 "accounting": "2 x 64 integer operations per (query, database vector) pair; device time of norms + search + merge, HIP events, mean of %d runs" % reps},
    "cpu_baseline": {"value": round(m * float(n) / cdt, 1), "unit": "pairs/s", "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
                     "sample": "oracle o3_knn64 (brute force, OpenMP) on the first %d queries against all %d vectors: %.2f s; same neighbours as the GPU: %s" % (m, n, cdt, same)},
}))
