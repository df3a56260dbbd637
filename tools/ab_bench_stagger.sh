#!/bin/bash
# tools/ab_bench_stagger.sh [passes=2] ["1 2"] -- the whole bench step (512^3, production schedule) under SIFT3D_TUNE_FUSED_STAGGER values,
# alternated on one box: ms per step, roofline.frac, the five launches
PASSES=${1:-2}; SET=${2:-"1 2"}
for pass in $(seq 1 $PASSES); do
  for stg in $SET; do
    python3 bench.py --steps 20 --warmup 5 --cpu-sample 64 --tune FUSED_STAGGER=$stg > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
    python3 - $pass $stg <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("pass %s stagger %s: ms_per_step %.3f  frac %.3f  launch %.4f ms  ceiling_frac %s  per launch %s" % (
    sys.argv[1], sys.argv[2], d["ms_per_step"], r["frac"], r.get("avg_launch_ms", 0) or 0, r.get("ceiling_frac"),
    [(k.get("taps"), k.get("alg_bytes_per_voxel"), round(k.get("avg_launch_ms", 0), 4)) for k in r.get("per_instantiation", [])]))
PY
  done
done
