"""CPU suite: the Z-slab driver (3d_sift_cuda_amd/zslab.py) with gloo, world size 2 and 3 (and 1 / 3 in-process plans).

The compute backend here is the oracle (test infrastructure); what is under test is the slab logic that the GPU
path shares: boundaries, halo widths, exchange, DoG halo repair, own-slice filtering, coarse-octave gather, order.
The sharded run must reproduce the serial oracle's validated extrema exactly.
"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleBackend:
    """numpy/oracle stand-in for HipBackend (CPU tensors; candidates kept as python lists)."""

    def __init__(self, oracle, torch, bands_first=False):
        self.o, self.torch = oracle, torch
        self.cands = []
        self.bands_first = bands_first

    # the windowed blur of the HIP backend (boundary bands first): only the output planes [z_lo, z_hi) are written
    def window_ok(self, shape, sigma):
        return self.bands_first

    def blur_dog_window(self, src, dst, dog, z_lo, z_hi, sigma):
        out = self.torch.from_numpy(self.o.blur(src.numpy(), sigma))
        dst[z_lo:z_hi].copy_(out[z_lo:z_hi])
        if dog is not None:
            dog[z_lo:z_hi].copy_(self.torch.from_numpy(self.o.dog(src.numpy(), out.numpy()))[z_lo:z_hi])

    def empty(self, shape):
        return self.torch.zeros(shape, dtype=self.torch.float32)

    def from_host(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr, np.float32).copy())

    def blur(self, src, dst, sigma):
        dst.copy_(self.torch.from_numpy(self.o.blur(src.numpy(), sigma)))

    def blur_dog(self, src, dst, dog, sigma):
        out = self.o.blur(src.numpy(), sigma)
        dst.copy_(self.torch.from_numpy(out))
        dog.copy_(self.torch.from_numpy(self.o.dog(src.numpy(), out)))

    def dog(self, a, b, out):
        if a.numel():
            out.copy_(self.torch.from_numpy(self.o.dog(a.numpy(), b.numpy())))

    def subsample(self, src, dst):
        dst.copy_(self.torch.from_numpy(self.o.subsample(src.numpy())))

    def reset(self):
        self.cands = []

    def extrema_append(self, dp, dc, dn, level_id, z_lo, z_hi):
        mins, maxs = self.o.detect3(dp.numpy(), dc.numpy(), dn.numpy())
        for is_max, lst in ((0, mins), (1, maxs)):
            for e in lst:
                if z_lo <= e["z"] < z_hi:
                    h = dp[int(e["z"]), int(e["y"]), int(e["x"])].item()
                    l = dn[int(e["z"]), int(e["y"]), int(e["x"])].item()
                    self.cands.append((level_id, is_max, int(e["z"]), int(e["y"]), int(e["x"]), float(e["value"]), h, l))

    def level_entry(self, img, dogc, nz_global, z_offset, sh, sc, sl, factor):
        return {"z_offset": z_offset, "nz_global": nz_global}

    def candidates(self, table):
        import _oracle
        out = np.zeros(len(self.cands), _oracle.CAND)
        for i, (lid, is_max, z, y, x, v, h, l) in enumerate(sorted(self.cands)):
            out[i] = (lid // 3, lid % 3 + 1, is_max, x, y, z + table[lid]["z_offset"], v, h, l)
        return out

    def before_exchange(self):
        pass

    def after_exchange(self):
        pass


def _same(got, want):
    assert len(got) == len(want), (len(got), len(want))
    for f in ("octave", "level", "is_max", "x", "y", "z"):
        assert (got[f] == want[f]).all(), f
    for f in ("value", "h_value", "l_value"):
        assert (got[f].view(np.uint32) == want[f].view(np.uint32)).all(), f


def test_slab_plan_geometry():
    zs = importlib.import_module("3d_sift_cuda_amd").__name__ and importlib.import_module("3d_sift_cuda_amd.zslab")
    p = zs.SlabPlan(512, 512, 512, 8)
    assert p.n_sharded == 2 and p.bounds == [0, 64, 128, 192, 256, 320, 384, 448, 512]
    assert p.slab(3, 1) == (96, 128) and p.input_range(0) == (0, 80) and p.input_range(7) == (432, 512)
    p = zs.SlabPlan(1024, 1024, 512, 4)
    assert p.n_sharded == 3 and p.slab(1, 2) == (32, 64)
    p = zs.SlabPlan(2048, 2048, 1024, 8)
    assert p.n_sharded == 3 and all(b % 8 == 0 for b in p.bounds)
    p = zs.SlabPlan(64, 64, 64, 8)      # too thin for 32-slice halos: nothing is sharded
    assert p.n_sharded == 0
    p = zs.SlabPlan(100, 90, 150, 2)    # odd sizes: boundary aligned to 2^K
    assert p.n_sharded >= 1 and p.bounds[1] % (1 << p.n_sharded) == 0
    extra0, extras, sig = zs.sigma_schedule(1.0)
    assert abs(extra0 - 1.5198684930801392) < 1e-12 and abs(extras[4] - 3.0900158882141113) < 1e-12


def _worker(rank, world, port, dims, seed, q, bands_first=False):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    pkg = importlib.import_module("3d_sift_cuda_amd")
    zs = importlib.import_module("3d_sift_cuda_amd.zslab")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        vol = pkg.synth_blobs(*dims, seed=seed)
        plan = zs.SlabPlan(dims[0], dims[1], dims[2], world)
        i0, i1 = plan.input_range(rank)
        dgroup = dist.new_group(ranks=list(range(world)), backend="gloo") if bands_first else None
        ex = zs.ZSlabExtractor(OracleBackend(_oracle.load(), torch, bands_first), plan, rank, dist, deferred_group=dgroup)
        ex.run(vol[i0:i1], i0)
        mine = ex.candidates()
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        if rank == 0:
            allc = np.concatenate(gathered)
            key = (allc["octave"].astype(np.int64) * 3 + allc["level"] - 1) * 2 + allc["is_max"]
            merged = allc[np.argsort(key, kind="stable")]
            q.put((plan.n_sharded, merged, ex.stats))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("dims,seed,world,bands", [((40, 36, 160), 3, 2, False), ((36, 40, 130), 9, 2, True), ((28, 24, 208), 5, 3, True),
                                                   ((24, 20, 272), 8, 4, False), ((24, 20, 272), 8, 4, True),
                                                   ((20, 16, 1100), 2, 2, False)])   # (the last: every octave sharded, none gathered)
def test_gloo_ranks_match_serial_oracle(oracle, built, dims, seed, world, bands):
    """World size 2, 3 and 4: from 3 on there are ranks with a neighbour on both sides (what every interior rank of an
    8-GPU run is).  The schedule under test exchanges 8 slices per level and defers the rest of the L1..L3 halos
    to two batches per octave (on a process group of its own when given one): what the subsample reads, completed at the octave's
    end, and what only patches reach, completed at the end of run().  bands: boundary bands first -- a rank filters
    the two bands its neighbours fetch, issues the exchange, filters its interior and only then completes the exchange; its
    own halo slices are never computed, they arrive."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, seed, q, bands)) for r in range(world)]
    for p in procs:
        p.start()
    n_sharded, merged, stats = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert n_sharded >= 1 and stats["exchanges"] >= 5
    # rank 0's traffic per sharded octave: five 8-slice exchanges on the critical path, two deferred batches -- the eight
    # slices of L3 the subsample reads beyond +- 8, and the 11 + 15 + 12 slices of L1, L2, L3 that only patches reach
    # (zslab.PATCH_REACH = 19 / 23 / 28 slices, round 5; before: one batch of 3 x 24): 40 against 46 slices
    assert stats["deferred_exchanges"] == 2 * n_sharded and stats["exchanges"] == 7 * n_sharded
    assert stats["deferred_bytes"] * 86 == stats["exchange_bytes"] * 46
    # bands first: every per-level halo (the 40 slices) was issued before the interior of its level was filtered
    assert stats["hidden_bytes"] == (stats["exchange_bytes"] - stats["deferred_bytes"] if bands else 0)
    want = oracle.candidates(built.synth_blobs(*dims, seed=seed))
    assert len(want) > 20
    _same(merged, want)


def test_single_rank_plan_is_the_serial_path(oracle, built):
    import torch
    zs = importlib.import_module("3d_sift_cuda_amd.zslab")
    dims = (48, 40, 36)
    vol = built.synth_blobs(*dims, seed=5)
    plan = zs.SlabPlan(*dims, 1)
    ex = zs.ZSlabExtractor(OracleBackend(oracle, torch), plan, 0, None)
    ex.run(vol, 0)
    _same(ex.candidates(), oracle.candidates(vol))


def test_placed_shifts_are_the_merge_by_group_order():
    """zslab.placed_shifts (where a rank's descriptor kernel stores its records in the shared list) against merge_by_group (the host
    merge of rounds 1 - 4): for random per-rank group counts, record i of rank r, group g lands exactly where the stable sort by
    group of the concatenated ranks puts it."""
    zs = importlib.import_module("3d_sift_cuda_amd.zslab")
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 5, 9):
        groups = 193
        counts = rng.integers(0, 6, (world, groups)) * (rng.random((world, groups)) < 0.3)
        parts, where = [], []
        for r in range(world):
            grp = np.repeat(np.arange(groups), counts[r])              # a rank's own records are sorted by group
            recs = np.zeros(len(grp), [("rank", "<i4"), ("i", "<i4")])
            recs["rank"], recs["i"] = r, np.arange(len(grp))
            parts.append((recs, grp.astype(np.int32)))
            shift, total = zs.placed_shifts(counts, r)
            where.append(np.arange(len(grp)) + shift[grp])
        assert total == counts.sum()
        merged = zs.merge_by_group(parts)
        if merged is None:
            assert total == 0
            continue
        placed = np.zeros(total, merged.dtype)
        seen = np.zeros(total, bool)
        for (recs, _), w in zip(parts, where):
            assert not seen[w].any()
            placed[w] = recs
            seen[w] = True
        assert seen.all() and placed.tobytes() == merged.tobytes()
