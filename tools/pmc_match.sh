#!/bin/bash
# Development aid (GPU box): PMC passes over `python3 tools/bench_match.py ARGS` (ARGS quoted as one word, e.g. "100 2000 5 1 1"), per-kernel averages for kernels
# matching PATTERN.  usage: bash tools/pmc_match.sh "<bench_match args>" <out-subdir> <pattern> <counter group> [...]
set -e
N=$1; OUT=$2; PAT=$3; shift 3
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do
  tag=$(echo "$grp" | tr ' ,' '__')
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp -d $ROOT/gpurun_out/$OUT/$tag -o pmc --output-format csv -- python3 $ROOT/tools/bench_match.py $N > $ROOT/gpurun_out/$OUT/$tag.log 2>&1 || { tail -5 $ROOT/gpurun_out/$OUT/$tag.log; exit 1; }
done
cd $ROOT
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob("gpurun_out/%s/*/**/*counter_collection.csv" % out, recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(agg.items()):
        print(k, {c: (round(sum(v) / len(v)), len(v)) for c, v in d.items()})
PY
