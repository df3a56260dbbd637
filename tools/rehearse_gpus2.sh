#!/bin/bash
# tools/rehearse_gpus2.sh RUNS SIZE [RANKS] -- `python3 bench.py --gpus RANKS` started WITHOUT a launcher on a one-GPU box, the
# ranks sharing the GPU over gloo, RUNS times in a row; stops at the first run that fails, stalls (bench.py's per-phase watchdog
# names the phase) or does not print the round-5 line: n_gpus RANKS, scaling "strong", value = the Z-slab split of ONE volume
# with the single-GPU bytes, the replica leg as volumes_value, and zslab_c (the C driver's resident form) with both transports --
# the RCCL one through tests/rccl_shim, because real RCCL refuses two ranks on one device.  Then (round 6) two more runs with a failure
# injected (--zslab-inject): the shared record list refused on rank 1, and the per-process job dead before its first collective.
RUNS=${1:-20}; SIZE=${2:-256}; RANKS=${3:-2}
OUT=gpurun_out/rehearse; mkdir -p $OUT
export SIFT3D_DIST_BACKEND=gloo
export SIFT3D_BENCH_RCCL_LIBRARY=$(pwd)/tests/rccl_shim/_build/librccl_shim.so
for i in $(seq 1 $RUNS); do
  t0=$(date +%s.%N)
  timeout -k 10 600 python3 bench.py --gpus $RANKS --steps 3 --warmup 1 --size $SIZE --cpu-sample 0 --phase-limit 120 \
      > $OUT/run_$i.json 2> $OUT/run_$i.err
  rc=$?
  t1=$(date +%s.%N)
  python3 - $OUT/run_$i.json $rc $i $t0 $t1 $RANKS <<'PY' || { echo "run $i FAILED (rc $rc)"; tail -40 $OUT/run_$i.err; exit 1; }
import json, sys
p, rc, i, t0, t1, ranks = sys.argv[1], int(sys.argv[2]), sys.argv[3], float(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
lines = [l for l in open(p) if l.startswith("{")]
ok = rc == 0 and len(lines) == 1
d = json.loads(lines[0]) if ok else {}
zc = d.get("zslab_c") or {}
pc, rc_ = zc.get("peer_copy") or {}, zc.get("rccl") or {}
ok = ok and d.get("n_gpus") == ranks and d.get("scaling") == "strong" and d.get("same_bytes_as_single_gpu") is True
ok = ok and d.get("value") == d["zslab"]["value"] and d.get("volumes_value") and d.get("volumes_scaling") == "weak"
ok = ok and pc.get("same_bytes_as_single_gpu") is True and rc_.get("same_bytes_as_single_gpu") is True
ok = ok and rc_.get("transport") == "rccl" and rc_.get("comm_sets") == 2 and pc.get("transport") == "peer_copy"
print("run %s: rc %d n_gpus %s scaling %s zslab ms %s (same bytes %s) volumes ms %s | zslab_c peer %s ms (same %s) rccl[%s] %s ms (same %s)  wall %.1f s" % (
    i, rc, d.get("n_gpus"), d.get("scaling"), d.get("ms_per_step"), d.get("same_bytes_as_single_gpu"), d.get("volumes_ms_per_step"),
    pc.get("ms_per_step"), pc.get("same_bytes_as_single_gpu"), rc_.get("transport"), rc_.get("ms_per_step"), rc_.get("same_bytes_as_single_gpu"), t1 - t0), flush=True)
sys.exit(0 if ok else 1)
PY
done | tee $OUT/summary.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
# Round 6: the two fall-backs of the N > 1 line, each with its failure injected.
#  shared-list: rank 1 cannot register the shared record list -> every rank gathers instead, same bytes, and the line says why;
#  torch-leg:   the per-process job dies before its first collective -> the line's value is the one-process C driver's (which ran
#               FIRST), value_source says so, exit code 0.
for inj in shared-list torch-leg; do
  timeout -k 10 600 python3 bench.py --gpus $RANKS --steps 2 --warmup 1 --size $SIZE --cpu-sample 0 --phase-limit 120 --zslab-inject $inj \
      > $OUT/inject_$inj.json 2> $OUT/inject_$inj.err
  rc=$?
  python3 - $OUT/inject_$inj.json $rc $inj $RANKS <<'PY' || { echo "injected $inj FAILED (rc $rc)"; tail -40 $OUT/inject_$inj.err; exit 1; }
import json, sys
p, rc, inj, ranks = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
lines = [l for l in open(p) if l.startswith("{")]
assert rc == 0 and len(lines) == 1, (rc, len(lines))
d = json.loads(lines[0])
z, zc = d["zslab"], d["zslab_c"]
assert d["n_gpus"] == ranks and d["scaling"] == "strong" and d["value"] and d["same_bytes_as_single_gpu"] is True
if inj == "shared-list":
    why = z["records_gathered_because"]   # rank 0's view: "... rank(s) [1] could not set it up" (rank 1's own message names the injection)
    assert z["status"] == "ok" and z["records_to_rank0"].startswith("gathered") and "shared record list" in why and ("[1]" in why or "injected" in why), z
    assert d["value_source"].startswith("zslab: one process per GPU")
    assert len(z["per_rank"]) == ranks and all(r["halo_bytes_critical"] > 0 and r["per_octave"] for r in z["per_rank"])
else:
    assert d["value_source"].startswith("zslab_c/peer_copy") and z["per_process_job"]["exit_code"] not in (0, None), d["value_source"]
    pc = zc["peer_copy"]
    assert abs(d["ms_per_step"] - pc["ms_per_step"]) < 1e-9 and pc["same_bytes_as_single_gpu"] is True
print("injected %s: rc %d value %.0f ms %.3f source: %s | records: %s" % (inj, rc, d["value"], d["ms_per_step"], d["value_source"][:60], str(z.get("records_to_rank0"))[:40]), flush=True)
PY
done | tee -a $OUT/summary.txt
exit ${PIPESTATUS[0]}
