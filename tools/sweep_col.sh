#!/bin/bash
# tuning aid: sweep the groups-per-chunk of the column blur kernel at 512^3
for k in 3 4 5 6 8 10 12 16 24 31; do echo "== SIFT3D_COL_K=$k"; SIFT3D_COL_K=$k python tools/bench_blur.py 512 10 2>&1 | grep -E "taps  7|taps 17"; done
