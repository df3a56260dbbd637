#!/usr/bin/env python3
"""GPU box: how many 128-byte lines of L4 the lazy level-above test (extrema_validate_lazy_kernel<8>) touches, counted exactly
from the extrema of the last detection level of octave 0 -- per candidate (what is fetched when no two candidates share a
line in cache) and as a union (what a perfect order of the candidates could at best bring it down to).  Round-4 review,
weak 4: "the builder's argument that the rest is line granularity is plausible and unmeasured".
A candidate at (x, y, z) reads the (2R+3)^3 = 19^3 block of L4 around it: rows [y-9, y+9] x planes [z-9, z+9], floats
[x-9, x+9] of each row (clipped to the volume).
usage: python tools/lazy_lines.py [n=512]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_sift_cuda_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol = pkg.synth_blobs(n, n, n, seed=12345)
with pkg.Context(n, n, n) as ctx:
    ctx.set_volume(vol)
    cand = ctx.detect()
c = cand[(cand["octave"] == 0) & (cand["level"] == 3)]
print("%d^3 blob field: %d validated extrema on octave 0, detection level 3 (the kernel's list also holds those the level above then refutes: about 22 000)" % (n, len(c)))
R = 9
LINE = 32   # floats per 128-byte line
per_cand = 0
lines = set()
rows_total = 0
for x, y, z in zip(c["x"].astype(np.int64), c["y"].astype(np.int64), c["z"].astype(np.int64)):
    x0, x1 = max(0, x - R), min(n - 1, x + R)
    l0, l1 = x0 // LINE, x1 // LINE
    ys = np.arange(max(0, y - R), min(n - 1, y + R) + 1)
    zs = np.arange(max(0, z - R), min(n - 1, z + R) + 1)
    nrows = len(ys) * len(zs)
    rows_total += nrows
    per_cand += nrows * (l1 - l0 + 1)
    row_id = (zs[:, None] * n + ys[None, :]).ravel() * (n // LINE)
    for l in range(l0, l1 + 1):
        lines.update((row_id + l).tolist())
level_bytes = 4.0 * n ** 3
print("rows read: %d (%.1f per candidate); lines per row: %.3f (a 76-byte segment starts anywhere in a 128-byte line: 1 + 18/32 = 1.5625)" % (rows_total, rows_total / len(c), per_cand / rows_total))
print("lines touched candidate by candidate (no sharing): %d = %.3f GB = %.2f x the level (%.2f GB); per candidate %.1f KB" % (per_cand, per_cand * 128 / 1e9, per_cand * 128 / level_bytes, level_bytes / 1e9, per_cand * 128 / len(c) / 1e3))
print("distinct lines (every line fetched once: the best any order of the candidates could do): %d = %.3f GB = %.2f x the level" % (len(lines), len(lines) * 128 / 1e9, len(lines) * 128 / level_bytes))
print("scaled to the kernel's ~22 000 list entries: no sharing %.2f GB, distinct about %.2f GB (distinct lines saturate: upper bound %.2f GB = the level)" % (per_cand * 128 / len(c) * 22000 / 1e9, min(level_bytes, len(lines) * 128 * 22000 / len(c)) / 1e9, level_bytes / 1e9))
