#!/bin/bash
# tools/ab_match_ahead.sh [pairs=3] -- the matcher's search kernel (development build) with the matrix cores ONE subtile ahead of the
# vector unit (the product's form) against TWO (three accumulator sets), alternated; 200 000 x 200 000 descriptors, k = 5, on the
# clustered data set and on uniformly random permutations.  Prints ms, TOP/s, fraction of the int8 peak, and whether the neighbours are
# the brute-force oracle's.
PAIRS=${1:-3}
for uniform in 0 1; do
  for p in $(seq 1 $PAIRS); do
    for ahead in 1 2; do
      KNN_AHEAD=$ahead python3 tools/bench_match.py 100 2000 5 5 $uniform 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('uniform $uniform pair $p ahead $ahead: %.3f ms  %.0f TOP/s  %.3f of peak  same as oracle: %s' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline']['sample'].split('GPU: ')[-1]))"
    done
  done
done
