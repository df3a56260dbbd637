#!/bin/bash
# GPU box: A/B of one tuning knob through bench.py, alternated in one process group per run.
#   bash tools/ab_knob.sh KNOB "v1 v2" "sizes" [steps] [pairs]     e.g.  bash tools/ab_knob.sh SPLIT_TAIL "1 0" "512 256 128" 20 2
KNOB=$1; VALS=$2; SIZES=$3; STEPS=${4:-20}; PAIRS=${5:-2}
for sz in $SIZES; do
  for p in $(seq $PAIRS); do
    for v in $VALS; do
      python3 bench.py --size $sz --cpu-sample 0 --steps $STEPS --tune $KNOB=$v 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('size %s  %s  %.3f ms per step  %.0f records/s' % (sys.argv[1], d['config'].get('tuning'), d['ms_per_step'], d['value']))" $sz
    done
  done
done
