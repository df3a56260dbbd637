#!/bin/bash
# GPU box: rocprofv3 kernel trace of a few extractions, then the extrema / validate launches grouped by (kernel, grid).
#   bash tools/trace_extrema.sh <tag> [bench args]   -> gpurun_out/trace_<tag>/dispatches.csv
set -e
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 "$@" > $OUT/bench.json 2> $OUT/err.log
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
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
PY
head -40 $OUT/dispatches.csv
