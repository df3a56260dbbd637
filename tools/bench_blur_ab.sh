#!/bin/bash
# A/B of the fused-blur forms (one process per variant: the launcher reads its switches once per call, the occupancy
# query is cached per instantiation).  usage: tools/bench_blur_ab.sh [N=512] [reps=10]
N=${1:-512}; REPS=${2:-10}
run() { echo "== $*"; env "$@" python tools/bench_blur.py $N $REPS 2>&1 | grep taps; }
run SIFT3D_RING_PRIO=0
run SIFT3D_RING_PRIO=1
run SIFT3D_RING_PRIO=0
run SIFT3D_RING_PRIO=1
