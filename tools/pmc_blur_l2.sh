#!/bin/bash
# GPU box: L2 behaviour of the fused blur instantiations at 512^3 (tools/bench_blur.py): hits / misses / requests of the
# TCC, bytes fetched through the fabric, per instantiation.  One rocprofv3 run per counter group (kernel trace + counters).
# usage: bash tools/pmc_blur_l2.sh <out-subdir>
OUT=$1
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i+1)); tag=g$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $ROOT/gpurun_out/$OUT/$tag -o pmc --output-format csv -- python3 $ROOT/tools/bench_blur.py 512 3 > $ROOT/gpurun_out/$OUT/$tag.log 2>&1 || { echo "group $tag failed: $grp"; tail -3 $ROOT/gpurun_out/$OUT/$tag.log; }
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/%s/*/**/*counter_collection.csv" % out, recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "blur_fused" not in k: continue
        k = k[k.index("blur_fused"):].split("(")[0][:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
N = 512 ** 3
for k, d in sorted(agg.items()):
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    print(k, {c: round(v) for c, v in sorted(med.items())})
    if "TCC_HIT_sum" in med and "TCC_MISS_sum" in med:
        print("   L2 hit rate %.3f; requests per voxel %.4f" % (med["TCC_HIT_sum"] / max(1.0, med["TCC_HIT_sum"] + med["TCC_MISS_sum"]), med.get("TCC_REQ_sum", 0) / N))
PY
