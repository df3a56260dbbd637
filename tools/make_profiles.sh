#!/bin/bash
# GPU box: regenerates the rocprofv3 evidence that profiles/ holds for one build.
#   bash tools/make_profiles.sh <tag>       (e.g. r01_c) -> gpurun_out/prof_<tag>/...
# 1. rocprofv3 --kernel-trace --stats over `python3 bench.py --steps 5 --warmup 2 --cpu-sample 0`
# 2. PMC passes (separate runs, kernel trace only): FETCH_SIZE, WRITE_SIZE over `python3 tools/bench_blur.py 512 3`
set -e
TAG=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o pmc --output-format csv -- python3 $ROOT/tools/bench_blur.py 512 3 > $OUT/pmc_$c.log 2>&1
done
cd $ROOT
python3 bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python3 - $TAG <<'PY'
import csv, glob, json, sys, collections, statistics
tag = sys.argv[1]
out = "gpurun_out/prof_%s" % tag
# per-kernel stats as rocprofv3 wrote them
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
if st:
    open(out + "/%s_kernel_stats.csv" % tag, "w").write(open(st[0]).read())
# dispatch groups of the blur kernels: (kernel, grid) -> n, min/avg/max us
tr = glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True)
if tr:
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        k = r["Kernel_Name"]
        if "blur" not in k and "extrema" not in k: continue
        k = k.split("(")[0].replace("void ", "")
        g[(k, r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out + "/%s_blur_dispatches.csv" % tag, "w", newline="") as f:
        w = csv.writer(f)   # kernel names hold commas
        w.writerow(["kernel", "grid", "workgroup", "calls", "min_us", "avg_us", "max_us"])
        for (k, gs, ws), v in sorted(g.items()):
            w.writerow([k, gs, ws, len(v), "%.1f" % min(v), "%.1f" % (sum(v) / len(v)), "%.1f" % max(v)])
# PMC traffic per blur kernel at 512^3
import hashlib
import datetime, platform, subprocess
def _gpu_name():
    # the GPU agent as rocprofv3 itself recorded it for the profiled run (name, CUs, clock); rocminfo's marketing name is
    # empty on some boxes of the pool
    try:
        f = glob.glob(out + "/stats/**/*agent_info.csv", recursive=True)[0]
        g = [r for r in csv.DictReader(open(f)) if r.get("Agent_Type") == "GPU"][0]
        return "%s, %s CUs, %s MHz" % (g.get("Name"), g.get("Cu_Count"), g.get("Max_Engine_Clk_Fcompute"))
    except Exception as e:
        return "unknown (%r)" % (e,)
res = {"_measured": {"date_utc": datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M"), "host": platform.node(), "gpu": _gpu_name(),
                     "by": "tools/make_profiles.sh " + tag},
       "_kernel_source_sha256": hashlib.sha256(open("3d_sift_cuda_amd/csrc/kernels_blur_fused.hip", "rb").read()).hexdigest(),
       "_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python3 tools/bench_blur.py 512 3` on MI355X; medians per launch at 512^3 (134 217 728 voxels). _kernel_source_sha256 is the hash of kernels_blur_fused.hip the passes ran on: bench.py drops `traffic` when the source has changed since. FETCH_SIZE (KB, TCC_EA0_RDREQ x 64 B) is doubled as MI355X_MICROARCH.md prescribes for 16-B/lane reads on gfx950; WRITE_SIZE (KB) is taken as is."}
vals = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c or "blur" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            vals[k][c] = statistics.median(v)
N = 512 ** 3
for k, d in sorted(vals.items()):
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d: continue
    rd, wr = d["FETCH_SIZE"] * 1024 * 2, d["WRITE_SIZE"] * 1024
    res[k] = {"FETCH_SIZE_KB_raw": d["FETCH_SIZE"], "WRITE_SIZE_KB": d["WRITE_SIZE"], "hbm_read_bytes_per_launch_512": rd,
              "hbm_write_bytes_per_launch_512": wr, "hbm_bytes_per_launch_512": rd + wr,
              "read_B_per_voxel": round(rd / N, 2), "write_B_per_voxel": round(wr / N, 2)}
json.dump(res, open(out + "/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(open(out + "/%s_kernel_stats.csv" % tag).read()[:3000] if st else "no stats")
PY
