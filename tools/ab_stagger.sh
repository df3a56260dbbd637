#!/bin/bash
# tools/ab_stagger.sh [N=512] [reps=20] [passes=3] -- the fused blur per instantiation with SIFT3D_TUNE_FUSED_STAGGER 1 (off: the kernel of
# rounds 2 - 5) and 2 (on: one copy of the march per wavefront role, the second half of the wavefronts half a step behind), alternated on
# one box.  (profiles/r06_stagger_ab.txt was taken with a development build that had two more modes and a prefetch knob.)
N=${1:-512}; REPS=${2:-20}; PASSES=${3:-3}
for pass in $(seq 1 $PASSES); do
  for stg in 1 2; do
    echo "== pass $pass stagger $stg"
    python3 tools/bench_blur.py $N $REPS 0 0 $stg 2>&1 | grep "^taps" | grep -v "taps 17"
  done
done
