#!/usr/bin/env python3
"""Agreement table between bench.py's HIP-event times and the rocprofv3 kernel trace of the same (profiled) run, for the
fused blur launches on the full-size volume -- what profiles/README.md quotes per round.
usage: python tools/agreement.py <prof dir written by tools/make_profiles.sh>   (e.g. gpurun_out/prof_r03)"""
import csv, glob, json, sys, collections
out = sys.argv[1]
bench = json.loads([l for l in open(out + "/bench_under_rocprof.json") if l.startswith("{")][-1])
per = {(p["taps"], p["alg_bytes_per_voxel"] > 9): p for p in bench["roofline"]["per_instantiation"]}
n = int(bench["config"]["workload"].split("^")[0])
tr = glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    k = r["Kernel_Name"]
    if "blur_fused_ring_kernel" not in k:
        continue
    k = k[k.index("blur_fused_ring_kernel"):].split("(")[0]
    g[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("| taps | B/voxel | kernel | HIP events, ms | rocprofv3 kernel trace, ms |")
print("|---|---|---|---|---|")
tot_ev = tot_tr = tot_bytes = 0.0
for (taps, dog), p in sorted(per.items()):
    R = taps // 2
    names = [k for k in g if k.startswith("blur_fused_ring_kernel<%d, 2," % R) and (", true, true," in k) == dog and (dog or ", true, false," in k)]
    if not names:
        continue
    v = sorted(g[names[0]], reverse=True)[:p["launches"] if "launches" in p else 9]   # the full-size launches are the longest of the instantiation
    nlaunch = min(len(v), 9)
    v = v[:nlaunch]
    ms_tr = sum(v) / len(v)
    bpv = p["alg_bytes_per_voxel"]   # 8 level only, 12 level + DoG, 12.5 when the launch also writes the next octave's level 0
    print("| %d | %g | `%s` | %.4f | %.4f |" % (taps, bpv, names[0], p["avg_launch_ms"], ms_tr))
    tot_ev += p["avg_launch_ms"]; tot_tr += ms_tr; tot_bytes += bpv * float(n) ** 3
print()
print("Sum of the %d launches: %.3f ms by HIP events, %.3f ms by the kernel trace (%.2f GB of compulsory bytes: %.2f / %.2f TB/s, "
      "%.3f / %.3f of 8 TB/s); the profiled run takes %.2f ms per step."
      % (len(per), tot_ev, tot_tr, tot_bytes / 1e9, tot_bytes / tot_ev / 1e9, tot_bytes / tot_tr / 1e9, tot_bytes / tot_ev / 1e9 / 8, tot_bytes / tot_tr / 1e9 / 8,
         bench["ms_per_step"]))
