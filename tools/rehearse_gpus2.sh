#!/bin/bash
# tools/rehearse_gpus2.sh RUNS SIZE -- `python3 bench.py --gpus 2` started WITHOUT a launcher on a one-GPU box, the two ranks
# sharing the GPU over gloo, RUNS times in a row; stops at the first run that fails, stalls (bench.py's per-phase watchdog
# names the phase) or does not report n_gpus 2 with the Z-slab records equal to the single-GPU bytes.
RUNS=${1:-20}; SIZE=${2:-256}
OUT=gpurun_out/rehearse; mkdir -p $OUT
export SIFT3D_DIST_BACKEND=gloo
for i in $(seq 1 $RUNS); do
  t0=$(date +%s.%N)
  timeout -k 10 400 python3 bench.py --gpus 2 --steps 3 --warmup 1 --size $SIZE --cpu-sample 0 --phase-limit 120 \
      > $OUT/run_$i.json 2> $OUT/run_$i.err
  rc=$?
  t1=$(date +%s.%N)
  python3 - $OUT/run_$i.json $rc $i $t0 $t1 <<'PY' || { echo "run $i FAILED (rc $rc)"; tail -40 $OUT/run_$i.err; exit 1; }
import json, sys
p, rc, i, t0, t1 = sys.argv[1], int(sys.argv[2]), sys.argv[3], float(sys.argv[4]), float(sys.argv[5])
lines = [l for l in open(p) if l.startswith("{")]
ok = rc == 0 and len(lines) == 1
d = json.loads(lines[0]) if ok else {}
ok = ok and d.get("n_gpus") == 2 and d.get("zslab_same_bytes_as_single_gpu") is True
print("run %s: rc %d n_gpus %s ms_per_step %s zslab_ms %s same_bytes %s  wall %.1f s" % (
    i, rc, d.get("n_gpus"), d.get("ms_per_step"), d.get("zslab_ms_per_step"), d.get("zslab_same_bytes_as_single_gpu"), t1 - t0), flush=True)
sys.exit(0 if ok else 1)
PY
done | tee $OUT/summary.txt
exit ${PIPESTATUS[0]}
