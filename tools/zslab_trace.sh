#!/bin/bash
# GPU box: the C slab driver with S slabs on one device, both settings of SIFT3D_ZSLAB_PATCH_WAIT untraced (wall time), then one
# rocprofv3 kernel + memory-copy trace of each, the LAST extraction of which is printed as a timeline (start, duration, what).
#   bash tools/zslab_trace.sh <tag> [S=2]
set -e
TAG=$1; S=${2:-2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/zslab_trace_$TAG
mkdir -p $OUT
for w in 0 1 0 1; do python3 tools/zslab_trace.py $S $w 20 >> $OUT/wall.txt; done
cat $OUT/wall.txt
cd /tmp && export TMPDIR=/tmp
for w in 0 1; do
  rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/w$w -o run --output-format csv -- python3 $ROOT/tools/zslab_trace.py $S $w 3 > $OUT/w$w.log 2> $OUT/w$w.err
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for w in (0, 1):
    ev = []
    for f in glob.glob(out + "/w%d/**/*kernel_trace.csv" % w, recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s grid %s" % (r.get("Queue_Id"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Grid_Size") or r.get("Grid_Size_X"))))
    for f in glob.glob(out + "/w%d/**/*memory_copy_trace.csv" % w, recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction"), r.get("Size") or r.get("Bytes") or "?")))
    ev.sort()
    # the last extraction: tools/zslab_trace.py sleeps 30 ms before it
    starts = [i for i in range(1, len(ev)) if ev[i][0] - max(e[1] for e in ev[max(0, i - 64):i]) > 10000000]
    i0 = starts[-1] if starts else 0
    with open(out + "/timeline_w%d.txt" % w, "w") as f:
        t0 = ev[i0][0]
        for s, e, what in ev[i0:]:
            f.write("%9.1f %8.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, what))
    print("patch_wait %d: %d events in the last extraction, span %.1f us" % (w, len(ev) - i0, (max(e[1] for e in ev[i0:]) - ev[i0][0]) / 1e3))
PY
