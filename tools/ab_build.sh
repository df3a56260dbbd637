#!/bin/bash
# tools/ab_build.sh [N=512] [reps=20] [passes=3] ["stg ..."=1 2 3] -- two builds of the library (csrc/_build against csrc/_build_ab, e.g. with
# and without the blur's max-ilp scheduling: make B=_build_ab BLUR_SCHED= _build_ab/libsift3d_hip.so), per fused-blur instantiation and
# stagger knob, alternated on one box
N=${1:-512}; REPS=${2:-20}; PASSES=${3:-3}; SET=${4:-"1 2 3"}
for pass in $(seq 1 $PASSES); do
  for b in _build _build_ab; do
    for stg in $SET; do
      echo "== pass $pass build $b stagger $stg"
      SIFT3D_AB_BUILD=$b python3 tools/bench_blur.py $N $REPS 0 0 $stg 2>&1 | grep "^taps" | grep -v "taps 17"
    done
  done
done
