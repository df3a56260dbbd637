#!/bin/bash
# A/B of the fused-blur forms (one process per variant: the launcher reads its switches once per call, the occupancy
# query is cached per instantiation).  usage: tools/bench_blur_ab.sh [N=512] [reps=10]
# Boxes of the pool differ by up to 15 % and drift by a few percent from run to run: compare within one call, and repeat
# the reference variant at the end.
N=${1:-512}; REPS=${2:-10}
run() { echo "== $*"; env "$@" python tools/bench_blur.py $N $REPS 2>&1 | grep taps; }
run SIFT3D_RING_BR=2                                   # default for 7-13 taps: two rows per thread, two planes ahead, one workgroup per CU
run SIFT3D_RING_BR=2 SIFT3D_RING_PF=1                  # two workgroups per CU, one plane ahead
run SIFT3D_RING_BR=1 SIFT3D_RING_PF=1                  # default for 17 taps: one row per thread, 1024 threads
run SIFT3D_RING_BR=1 SIFT3D_RING_PF=1 SIFT3D_RING_XO=4 # 4 outputs per x-pass lane
run SIFT3D_RING_PRIO=1                                 # issue priority for the x-pass wavefronts
run SIFT3D_FUSED_V=1                                   # the first form (round 1)
run SIFT3D_BLUR_FUSED=0                                # three launches per level
run SIFT3D_RING_BR=2
