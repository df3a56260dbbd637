"""Development aid (GPU box): shows that tests/test_gpu_parity.py::test_dev_entry_points_are_ordered_with_the_default_stream detects a
missing fence.  The same producer / *_dev call / consumer sequence runs once with the context on its own stream (fenced
against the default stream) and once with the context moved to a NON-BLOCKING stream created here (hipStreamNonBlocking,
like the context's own: a torch.cuda.Stream() is a blocking stream and is ordered with the default stream implicitly), where
nothing orders it against the default-stream producer."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
pkg = importlib.import_module("3d_sift_cuda_amd"); import _oracle
orc = _oracle.load()
dims = (96, 80, 72); nx, ny, nz = dims; sigma = 1.9465880393981934
base = pkg.synth_blobs(*dims, seed=8)
want = {k: orc.blur((base * np.float32(1 + k) + np.float32(k)).astype(np.float32), sigma) for k in range(4)}
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
for mode in ("fenced (context on its own stream)", "unfenced (context on a caller-made non-blocking stream)"):
    bad = 0
    with pkg.Context(*dims) as ctx:
        if mode.startswith("unfenced"):
            side = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(side), 1) == 0   # hipStreamNonBlocking
            ctx.set_stream(side.value)
        d_base = torch.from_numpy(base).cuda(); d_in = torch.empty_like(d_base); d_out = torch.empty_like(d_base); d_dog = torch.empty_like(d_base)
        junk = torch.empty((64, 1024, 1024), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
        for rnd in range(3):
            for k in range(4):
                for _ in range(3): junk.normal_()
                d_in.copy_(d_base * float(1 + k) + float(k)); d_out.fill_(float("nan"))
                ctx.gauss_blur_dog_dev(d_in.data_ptr(), d_out.data_ptr(), d_dog.data_ptr(), nx, ny, nz, sigma)
                got = d_out.clone(); d_in.fill_(-1.0)
                bad += int(not (got.cpu().numpy().view(np.uint32) == want[k].view(np.uint32)).all())
                torch.cuda.synchronize()
    print("%-58s wrong results in %d of 12 calls" % (mode, bad))
    # the pattern that failed in round 1: a tiny volume, the output cleared on the default stream right before the call
    sd = (256, 8, 8); sv = pkg.synth_blobs(*sd, seed=5) - np.float32(1.5); sw = orc.blur(sv, 1.5198684930801392); bad = 0
    with pkg.Context(*sd) as ctx:
        if mode.startswith("unfenced"):
            side = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(side), 1) == 0; ctx.set_stream(side.value)
        d_in = torch.from_numpy(sv).cuda(); d_out = torch.empty_like(d_in); torch.cuda.synchronize()
        for it in range(200):
            d_out.zero_()
            ctx.gauss_blur_dev(d_in.data_ptr(), d_out.data_ptr(), sd[0], sd[1], sd[2], 1.5198684930801392)
            ctx.sync()
            bad += int(not (d_out.cpu().numpy().view(np.uint32) == sw.view(np.uint32)).all())
    print("%-58s tiny volume, output cleared just before the call: wrong in %d of 200" % (mode, bad))
