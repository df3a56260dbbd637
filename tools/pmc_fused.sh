#!/bin/bash
# Development aid (GPU box): PMC passes over tools/bench_blur.py for the fused blur kernel.
# usage: bash tools/pmc_fused.sh <N> <out-subdir> <counter> [<counter> ...]   (one pass per counter group)
set -e
N=$1; OUT=$2; shift 2
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do
  tag=$(echo "$grp" | tr ' ,' '__')
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp -d $ROOT/gpurun_out/$OUT/$tag -o pmc --output-format csv -- python3 $ROOT/tools/bench_blur.py $N 2 > $ROOT/gpurun_out/$OUT/$tag.log 2>&1 || { tail -5 $ROOT/gpurun_out/$OUT/$tag.log; exit 1; }
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/%s/*/**/*counter_collection.csv" % out, recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "blur_fused" not in k: continue
        k = k[k.index("blur_fused"):][:28]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(agg.items()):
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
