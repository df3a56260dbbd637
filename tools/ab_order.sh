#!/bin/bash
# tools/ab_order.sh [N=512] [reps=20] -- the fused blur per instantiation under the three tile orders (SIFT3D_TUNE_FUSED_ORDER) and
# the two tiles, alternated twice on one box
N=${1:-512}; REPS=${2:-20}
for pass in 1 2; do
  for tile in 1 2; do
    for order in 1 2 3; do
      echo "== pass $pass tile $tile order $order"
      python3 tools/bench_blur.py $N $REPS $tile $order 2>&1 | grep "^taps" | grep -v "taps 17"
    done
  done
done
