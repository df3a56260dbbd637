#!/usr/bin/env python3
"""GPU box rehearsal of BASELINE config C4: the 1024 x 1024 x 512 volume cut into 4 Z-slabs, four processes sharing the one
GPU of the box, halos staged through the host (gloo).  Checks that the merged records are the bytes of the single-GPU
extraction.  (On a 4-GPU node the same driver runs with backend nccl; bench.py attaches that run.)
launch: SIFT3D_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
        --master-port 29551 tools/zslab_c4_gloo.py [NX NY NZ]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
pkg = importlib.import_module("3d_sift_cuda_amd")
zs = importlib.import_module("3d_sift_cuda_amd.zslab")
dims = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 1024, 512)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
vol = pkg.synth_blobs(*dims, seed=12345)
plan = zs.SlabPlan(dims[0], dims[1], dims[2], world)
i0, i1 = plan.input_range(rank)
ctx = pkg.Context(dims[0], dims[1], zs.slab_context_slices(plan, rank), device=0, slab=True)
be = zs.HipBackend(pkg, ctx, torch)
slab = torch.from_numpy(vol[i0:i1].copy()).cuda()
torch.cuda.synchronize()
dist.barrier()
t0 = time.perf_counter()
with be.stream_scope():
    ex = zs.ZSlabExtractor(be, plan, rank, dist)
    ex.run(slab, i0)
    recs, grp = ex.describe(desc_mode=0, copy=False)
    merged = zs.gather_records(dist, rank, world, recs, grp, "cuda:0", dtype=pkg.FEATURE_DTYPE)
dist.barrier()
dt = time.perf_counter() - t0
ctx.close()
if rank == 0:
    print("zslab %dx%dx%d over %d ranks (gloo, one GPU): %d records in %.2f s, sharded octaves %d, %d exchanges, %.0f MB of halos per rank"
          % (dims + (world, len(merged), dt, plan.n_sharded, ex.stats["exchanges"], ex.stats["exchange_bytes"] / 1e6)), flush=True)
    del slab
    torch.cuda.empty_cache()
    with pkg.Context(*dims) as c1:
        c1.set_volume(vol)
        want = c1.extract()
    same = len(want) == len(merged) and bool((merged.view(np.uint8) == want.view(np.uint8)).all())
    print("single GPU: %d records; merged Z-slab records identical bytes: %s" % (len(want), same), flush=True)
dist.barrier()
dist.destroy_process_group()
