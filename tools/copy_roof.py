#!/usr/bin/env python3
"""Development aid (GPU box): device-to-device copy rate of independent, tuned implementations, as a check on
tools/stream_roof.hip's own copy kernel: torch's elementwise copy, torch's clone, and the runtime's hipMemcpyAsync
(device to device), at several sizes.  GB/s counts bytes read + bytes written."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
for mb in (256, 512, 1024, 2048, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    res = {}
    for name, fn in (("torch copy_", lambda: b.copy_(a)),
                     ("torch add (1 read + 1 write)", lambda: torch.add(a, 1.0, out=b)),
                     ("hipMemcpyAsync d2d", lambda: hip.hipMemcpyAsync(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(n * 4), 3, ctypes.c_void_p(0)))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[name] = 2 * n * 4 / ts[len(ts) // 2] / 1e6
    print("%5d MB per array: " % mb + "  ".join("%s %.0f GB/s" % kv for kv in res.items()), flush=True)
