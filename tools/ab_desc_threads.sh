#!/bin/bash
# tools/ab_desc_threads.sh [N=512] -- the SIFT-rank descriptor kernel with one wavefront and 5.5 KB of LDS per record (SIFT3D_TUNE_DESC_THREADS 64)
# against two wavefronts and 10.6 KB (128), by sampling-token count, alternated; step time from bench.py, kernel time from its stages.
N=${1:-512}
for pass in 1 2; do
  for cfg in "DESC_THREADS=128 SAMPLER_CAP=4" "DESC_THREADS=64 SAMPLER_CAP=4" "DESC_THREADS=64 SAMPLER_CAP=6" "DESC_THREADS=64 SAMPLER_CAP=8" "DESC_THREADS=64 SAMPLER_CAP=12" "DESC_THREADS=64 SAMPLER_CAP=0"; do
    args=""; for kv in $cfg; do args="$args --tune $kv"; done
    python3 bench.py --steps 20 --warmup 3 --size $N --cpu-sample 0 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('pass $pass  %-40s step %.3f ms  descriptor kernel %.3f ms (exclusive %.3f)  keypoint kernel %.3f ms  records %d' % ('$cfg', d['ms_per_step'], d['stages']['descriptor']['ms_per_step'], d['stages']['descriptor'].get('exclusive_ms_per_step', 0), d['stages']['keypoint']['ms_per_step'], d['config']['records_per_volume']))"
  done
done
