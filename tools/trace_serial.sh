#!/bin/bash
# GPU box: rocprofv3 kernel trace of tools/serial_extract.py (every launch alone on the chip), launches grouped by (kernel, grid);
# with a second argument, one more pass per listed PMC counter group (comma-separated groups of space-separated counters).
#   bash tools/trace_serial.sh <tag> ["SQ_WAVES SQ_INSTS_VALU,FETCH_SIZE"] [kernel regex]
set -e
TAG=$1; PMC=$2; KRE=${3:-lazy}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/serial_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $ROOT/tools/serial_extract.py 512 3 > $OUT/run.log 2> $OUT/err.log
IFS=',' read -ra PMCG <<< "$PMC"
gi=0
for g in "${PMCG[@]}"; do
  rocprofv3 --kernel-trace --pmc $g -d $OUT/pmc_$gi -o pmc --output-format csv -- python3 $ROOT/tools/serial_extract.py 512 1 > $OUT/pmc_$gi.log 2>&1
  gi=$((gi+1))
done
cd $ROOT
python3 - $OUT "$KRE" <<'PY'
import csv, glob, sys, collections, re
out, kre = sys.argv[1], sys.argv[2]
tr = glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True)
g = collections.defaultdict(list)
for r in csv.DictReader(open(tr[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    g[(k, r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out + "/dispatches.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "workgroup", "calls", "min_us", "avg_us", "max_us", "total_us"])
    for (k, gs, ws), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k[:110], gs, ws, len(v), "%.1f" % min(v), "%.1f" % (sum(v) / len(v)), "%.1f" % max(v), "%.0f" % sum(v)])
print(open(out + "/dispatches.csv").read()[:6000])
for d in sorted(glob.glob(out + "/pmc_*/")):
    for fcsv in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(fcsv)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not re.search(kre, k): continue
            key = (k[:60], r.get("Grid_Size"))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        for key, cs in acc.items():
            print(key, dict(cs))
PY
