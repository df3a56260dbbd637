#!/usr/bin/env python3
"""GPU box: the zero-arithmetic march of the fused blur's tiles (tools/roof_lib.hip) by PLACEMENT of its three buffers -- back to
back (three 2^29-byte buffers: the three streams of a workgroup on the same channels at the same moment) or shifted by odd amounts.
usage: python tools/roof_placement.py [reps=10] [runs=3]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libroof.so"))
L.roof_march_detail.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names = ["64x32, 1 store", "128x16, 1 store", "64x32, 2 stores", "128x16, 2 stores"]
pads = ["back to back", "+ 1.04 MB / + 2.07 MB", "+ 12.7 MB / + 25.5 MB"]
for run in range(runs):
    best, det = (ctypes.c_float * 4)(), (ctypes.c_float * 12)()
    assert L.roof_march_detail(512, reps, best, det) == 0
    print("run %d (best of %d launches, ms; 512^3)" % (run, reps))
    for p in range(3):
        print("  %-24s " % pads[p] + "  ".join("%s %.4f" % (names[i], det[p * 4 + i]) for i in range(4)))
