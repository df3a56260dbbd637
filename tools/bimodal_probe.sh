#!/bin/bash
# tools/bimodal_probe.sh [procs=6] -- the 9-tap + DoG launch of the pyramid (the one whose time varies by process: 0.287 - 0.335 ms in the
# bench lines of round 6) in PROCS fresh processes per configuration: tile 1 = 64 x 32 / 2 = 128 x 16, stagger 1 = off / 2 = on; blob-field
# input as in bench.py.  One line per process: median of 20 launches.
PROCS=${1:-6}
for cfg in "2 1" "1 1" "2 2" "1 2"; do
  set -- $cfg
  for p in $(seq 1 $PROCS); do
    python3 - $1 $2 <<'PY'
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("3d_sift_cuda_amd")
tile, stg = int(sys.argv[1]), int(sys.argv[2])
n = 512
ctx = pkg.Context(n, n, n)
ctx.set_tuning(pkg.TUNE_FUSED_TILE, tile); ctx.set_tuning(pkg.TUNE_FUSED_STAGGER, stg)
a = torch.from_numpy(pkg.synth_blobs(n, n, n, seed=12345)).cuda()
b = torch.empty_like(a); d = torch.empty_like(a)
torch.cuda.synchronize()
out = []
for s, name in ((1.5450079441070557, "9+DoG"), (1.2262736558914185, "7+DoG")):
    for _ in range(3): ctx.gauss_blur_dog_dev(a.data_ptr(), b.data_ptr(), d.data_ptr(), n, n, n, s)
    ctx.enable_timing(True)
    for _ in range(20): ctx.gauss_blur_dog_dev(a.data_ptr(), b.data_ptr(), d.data_ptr(), n, n, n, s)
    log = ctx.launch_log(); ctx.enable_timing(False)
    ms = np.sort(log[log["stage"] == 7]["ms"])
    out.append("%s median %.4f min %.4f max %.4f" % (name, float(np.median(ms)), float(ms[0]), float(ms[-1])))
print("tile %d stagger %d: %s" % (tile, stg, " | ".join(out)), flush=True)
PY
  done
done
