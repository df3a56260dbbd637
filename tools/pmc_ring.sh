#!/bin/bash
# Development aid (GPU box): PMC passes over tools/bench_blur.py, one rocprofv3 run per counter group (kernel trace +
# counters only), continuing past a group whose counter names this rocprofv3 does not know.
# usage: bash tools/pmc_ring.sh <N> <out-subdir> [kernel-name filter = blur_fused] [program + args = tools/bench_blur.py <N> 2]
N=$1; OUT=$2; FILTER=${3:-blur_fused}
shift 3 2>/dev/null || shift $#
if [ $# -gt 0 ]; then PROG="$*"; else PROG="tools/bench_blur.py $N 2"; fi
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); tag=g$i
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp -d $ROOT/gpurun_out/$OUT/$tag -o pmc --output-format csv -- python3 $ROOT/$PROG > $ROOT/gpurun_out/$OUT/$tag.log 2>&1 || { echo "group $tag failed: $grp"; tail -3 $ROOT/gpurun_out/$OUT/$tag.log; }
done
cd $ROOT
python3 - "$OUT" "$FILTER" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
flt = sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/%s/*/**/*counter_collection.csv" % out, recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k: continue
        k = k[k.index(flt):].split("(")[0][:72]
        agg[k + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("gpurun_out/%s/g1/**/*kernel_trace.csv" % out, recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k: continue
        k = k[k.index(flt):].split("(")[0][:72] + " grid=" + (r.get("Grid_Size") or r.get("Grid_Size_X") or "?")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in sorted(agg.items()):
    print(k, "us=%.0f" % (sum(dur[k]) / max(1, len(dur[k]))), {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
