"""Z-slab multi-GPU driver of the extraction path (new work, the reference is single-GPU).

One process per GPU; the volume is cut along z (the slowest axis, so a slab is one contiguous block) into one
slab per rank.  Every level buffer holds the slab plus halos; after each Gaussian level the halo slices are
refreshed from the two neighbours with point-to-point ``torch.distributed`` sends/receives (backend "nccl" = RCCL
over xGMI on the GPUs; a slab chain only ever talks to its two neighbours, i.e. 2 of a GPU's 7 links).  There is
no other data-path collective until the coarse octaves (slab thinner than the halo) are gathered onto rank 0 and
the records are gathered at the end.

Why the result equals the single-GPU result bit for bit:
  * a blur output depends on inputs within R <= 8 slices; every level is (re)computed on slab +- 8 slices from an
    input that is exact there, the rows nearer than R to the artificial edge are overwritten by the neighbour's
    exact values, and at a face of the whole volume the computed region ends at the face, so the zero border is
    the reference's;
  * 2:1 subsampling pairs slices (2k, 2k+1): slab boundaries are multiples of 2^K (K = number of sharded octaves);
  * extrema are kept only for a rank's own slices (the halo slices are the neighbour's); lists sort by
    (level, is_max, index) and slabs are in z order, so concatenating ranks per group is the serial raster order;
  * per-keypoint geometry is computed in whole-volume coordinates (sift3d_level_desc.z_offset).

Halo widths: 8 slices feed the next blur (R <= 8); the buffers of L1..L3 keep 32 slices around a slab, of which the patches
of the slab's own keypoints can reach 19 / 23 / 28 (PATCH_REACH below: |offset| <= 5*sqrt(3) samples * 2*scale/5, scale <=
sigma_c + sigma_l of the detection level); DoG levels need 1.

Exchange schedule.  What the next blur waits for is only the 8-slice halo, so that is all a level exchanges on the critical
path: four or five exchanges of 8 slices per octave and direction (round 2).  The rest of the halos of L1..L3 is needed by
nothing before the per-keypoint stage, except the eight slices of L3 beyond +- 8 that the subsample seeding the next octave
reads (slab +- 16).  Round 5 fetches them in TWO batches issued as soon as L3 is complete: those eight slices of L3, waited for
at the end of the octave; and the patch-only slices (11 + 15 + 12 per direction), waited for when the whole pyramid has been
queued -- so on RCCL (which runs its transfers on a stream of its own) they have every later octave to arrive in.  Rounds 2 - 4
moved 3 x 24 slices in one batch and made the next octave wait for all of it; round 1 exchanged 32 slices after each of L1..L3
and waited each time.  Per octave and direction: 33 (40 with every level stored) slices on the critical path, 8 at the octave's
end, 38 at the run's end.

The compute backend is pluggable: ``HipBackend`` drives the C-ABI ``*_dev`` operators on torch CUDA tensors;
the CPU test-suite plugs the oracle in (tests/test_zslab_cpu.py) to check the slab logic with gloo, world size 2.
"""
import math

import numpy as np

HALO = 32        # slices of L1..L3 a slab's buffers keep around it (and the least a slab must be thick)
# Slices of L1, L2, L3 beyond a slab that the patches of its own keypoints can reach (round 5; before: HALO of each).  A keypoint of
# detection level l samples L_l within 5 sqrt(3) steps of 2 s / 5 of its centre; its scale s is twice the vertex of the parabola
# through the three sigmas, and the centre DoG value being a strict extremum of the three puts that vertex between the midpoints
# of the two intervals: s <= sigma_c + sigma_l = 4.556, 5.740, 7.232.  With the centre's refinement and the trilinear footprint:
# 18, 22, 27 slices; one more each for rounding.  (csrc/zslab_driver.hip: ZS_PATCH_REACH, and its NaN-poison test knob.)
PATCH_REACH = (None, 19, 23, 28)
SUB_HALO = 16    # the subsample that seeds the next octave's slab +- 8 reads L3 on slab +- 16
BLUR_HALO = 8    # slices recomputed / exchanged for the next blur (largest half-width in the schedule)


def sigma_schedule(initial_image_scale=1.0):
    """Float32 sigma arithmetic of msGeneratePyramidDOG3D_efficient (MultiScale.cpp:288-294,369,526-527)."""
    f32 = np.float32
    sigma_init = f32(0.5)
    if initial_image_scale > 0:
        sigma_init = f32(sigma_init / f32(initial_image_scale))
    s0 = f32(1.6)
    factor = f32(math.pow(2.0, 1.0 / 3.0))
    extra0 = f32(np.sqrt(f32(f32(s0 * s0) - f32(sigma_init * sigma_init))))
    extras, sig = [], [s0]
    s = s0
    for _ in range(5):
        extras.append(f32(s * f32(np.sqrt(f32(f32(factor * factor) - f32(1.0))))))
        s = f32(s * factor)
        sig.append(s)
    return float(extra0), [float(e) for e in extras], [float(v) for v in sig]


class SlabPlan:
    """Pure geometry: octave sizes, slab boundaries, how many octaves are sharded."""

    def __init__(self, nx, ny, nz, nranks):
        self.nx, self.ny, self.nz, self.nranks = nx, ny, nz, nranks
        self.octaves = []
        x, y, z = nx, ny, nz
        while x > 2 and y > 2 and z > 2 and len(self.octaves) < 32:
            self.octaves.append((x, y, z))
            x, y, z = x // 2, y // 2, z // 2
        # K = number of sharded octaves: boundaries multiples of 2^K, every slab at octave K-1 at least HALO thick
        self.n_sharded = 0
        self.bounds = [0, nz]
        if nranks > 1:
            for k in range(len(self.octaves), 0, -1):
                align = 1 << k
                b = [int(round(r * nz / nranks / align)) * align for r in range(nranks)] + [nz]
                ok = all(b[r + 1] > b[r] for r in range(nranks))
                for o in range(k):
                    zo = self.octaves[o][2]
                    for r in range(nranks):
                        lo, hi = b[r] >> o, (zo if r == nranks - 1 else b[r + 1] >> o)
                        ok = ok and (hi - lo) >= HALO
                if ok:
                    self.n_sharded, self.bounds = k, b
                    break
            if self.n_sharded == 0:
                self.bounds = [0] * nranks + [nz]   # too thin to shard: rank nranks-1... (see slab())
        else:
            self.n_sharded = 0

    def slab(self, rank, octave):
        """Global z-range [z0, z1) of `rank` in `octave` (a sharded octave)."""
        zo = self.octaves[octave][2]
        z0 = self.bounds[rank] >> octave
        z1 = zo if rank == self.nranks - 1 else self.bounds[rank + 1] >> octave
        return z0, z1

    def input_range(self, rank):
        """Slices of the input volume rank needs: its slab of octave 0 plus BLUR_HALO + the initial blur's reach."""
        if self.n_sharded == 0:
            return (0, self.nz)
        z0, z1 = self.slab(rank, 0)
        h = 2 * BLUR_HALO
        return (max(0, z0 - h), min(self.nz, z1 + h))


def slab_context_slices(plan, rank):
    """nz_local for sift3d_create_slab on `rank`: the most slices of an nx * ny volume one call is handed -- the input slab
    of the initial blur, a level buffer with its patch halos, and on rank 0 the first unsharded octave, which is gathered
    there (a whole octave, but of planes a 4^K-th the size)."""
    i0, i1 = plan.input_range(rank)
    need = (i1 - i0) + 2 * HALO
    K = plan.n_sharded
    if rank == 0 and 0 < K < len(plan.octaves):
        X, Y, Z = plan.octaves[K]
        pitch = lambda x: (x + 3) // 4 * 4
        plane = pitch(plan.nx) * plan.ny
        need = max(need, (pitch(X) * Y * Z + plane - 1) // plane)
    return need


def coarse_octave_dims(plan):
    """(nx, ny, nz) of the first octave that is not sharded -- the size of the second context rank 0 may give the octaves below the
    sharded ones (ZSlabExtractor(coarse_backend=...)) -- or None when there is nothing to give it."""
    if 0 < plan.n_sharded < len(plan.octaves):
        return tuple(plan.octaves[plan.n_sharded])
    return None


class HipBackend:
    """Compute on torch CUDA tensors through the C-ABI *_dev operators."""

    def __init__(self, pkg, ctx, torch, lazy_levels=True, bands_first=True):
        self.pkg, self.ctx, self.torch = pkg, ctx, torch
        self.lazy_levels = lazy_levels   # False: every level stored and filtered in full (the reference's schedule)
        self.bands_first = bands_first   # False: a level in one piece, then its exchange (the round-2 schedule)
        # one non-default torch stream carries everything: torch allocations/copies, the library's kernels
        # (sift3d_set_stream) and the point where NCCL work is ordered against (its current stream)
        self.stream = torch.cuda.Stream(device=ctx.device)
        ctx.set_stream(self.stream.cuda_stream)

    def stream_scope(self):
        return self.torch.cuda.stream(self.stream)

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device="cuda:%d" % self.ctx.device)

    def from_host(self, arr):
        if self.torch.is_tensor(arr):   # already resident on the device (the benchmark keeps its slab in HBM)
            return arr
        return self.torch.from_numpy(np.ascontiguousarray(arr, np.float32)).to("cuda:%d" % self.ctx.device)

    def blur(self, src, dst, sigma):
        nz, ny, nx = src.shape
        self.ctx.gauss_blur_dev(src.data_ptr(), dst.data_ptr(), nx, ny, nz, sigma)

    def blur_dog(self, src, dst, dog, sigma):
        nz, ny, nx = src.shape
        self.ctx.gauss_blur_dog_dev(src.data_ptr(), dst.data_ptr(), dog.data_ptr(), nx, ny, nz, sigma)

    def dog(self, a, b, out):
        self.ctx.dog_dev(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel())

    # ---- the blur restricted to output planes [z_lo, z_hi) of the buffer: optional part of the backend interface ----
    def window_ok(self, shape, sigma):
        nz, ny, nx = shape
        return self.bands_first and self.ctx.blur_window_supported(nx, ny, sigma)

    def blur_dog_window(self, src, dst, dog, z_lo, z_hi, sigma):
        nz, ny, nx = src.shape
        self.ctx.gauss_blur_dog_window_dev(src.data_ptr(), dst.data_ptr(), dog.data_ptr() if dog is not None else 0, nx, ny, nz,
                                           z_lo, z_hi, sigma)

    def subsample(self, src, dst):
        nz, ny, nx = src.shape
        self.ctx.subsample2_dev(src.data_ptr(), nx, ny, nz, dst.data_ptr())

    def reset(self):
        self.ctx.candidates_reset()

    def extrema_append(self, dp, dc, dn, level_id, z_lo, z_hi):
        nz, ny, nx = dc.shape
        self.ctx.extrema_append_dev(dp.data_ptr(), dc.data_ptr(), dn.data_ptr(), nx, ny, nz, level_id, z_lo, z_hi)

    # ---- neighbour levels that are not stored: optional part of the backend interface ----
    def lazy_ok(self, shape, next_sigma):
        nz, ny, nx = shape
        return self.lazy_levels and self.ctx.lazy_levels_supported(nx, ny, nz, next_sigma)

    def extrema_append_pair(self, ga, gb, dc, dn, level_id, z_lo, z_hi):
        """the level below is ga - gb"""
        nz, ny, nx = dc.shape
        self.ctx.extrema_append_lazy_dev(0, ga.data_ptr(), gb.data_ptr(), dc.data_ptr(), dn.data_ptr(), 0, 0.0, nx, ny, nz,
                                         level_id, z_lo, z_hi)

    def extrema_append_lazy_next(self, dp, dc, g, next_sigma, level_id, z_lo, z_hi):
        """the level above is g - blur(g, next_sigma), evaluated around the candidates only"""
        nz, ny, nx = dc.shape
        self.ctx.extrema_append_lazy_dev(dp.data_ptr(), 0, 0, dc.data_ptr(), 0, g.data_ptr(), next_sigma, nx, ny, nz, level_id,
                                         z_lo, z_hi)

    def level_entry(self, img, dogc, nz_global, z_offset, sh, sc, sl, factor):
        nz, ny, nx = img.shape
        return {"img": img.data_ptr(), "dogc": dogc.data_ptr(), "nx": nx, "ny": ny, "nz_local": nz, "nz_global": nz_global,
                "z_offset": z_offset, "sigma_h": sh, "sigma_c": sc, "sigma_l": sl, "octave_factor": factor, "_keep": (img, dogc)}

    def candidates(self, levels):
        return self.ctx.candidates_dev(levels)

    def describe(self, levels, desc_mode, eig_thres, size_factor, copy=True):
        return self.ctx.describe_dev(levels, desc_mode, eig_thres, size_factor, copy=copy)

    def describe_counts(self, levels, desc_mode, eig_thres, size_factor):
        return self.ctx.describe_dev_counts(levels, desc_mode, eig_thres, size_factor)

    def describe_place(self, list_address, shift):
        return self.ctx.describe_dev_place(list_address, shift)

    def before_exchange(self):
        pass  # the context runs on torch's current stream (ZSlabExtractor sets it), so ordering is the stream's

    def after_exchange(self):
        pass


class ZSlabExtractor:
    """Runs the pyramid of ONE volume across the ranks of a torch.distributed group."""

    def __init__(self, backend, plan, rank, dist=None, group=None, deferred_group=None, poison_halo=False, coarse_backend=None):
        """deferred_group: a second process group over the same ranks (dist.new_group()) for the once-per-octave batch of
        deferred patch halos.  RCCL serialises the operations of one communicator on one internal stream, so on the main
        group that batch (3 x 24 slices) would sit in front of the next level's 8-slice exchange; a group of its own has its
        own communicator and stream.  None: the main group carries both (correct, but the deferred bytes then delay the
        critical ones)."""
        self.be, self.plan, self.rank, self.dist, self.group = backend, plan, rank, dist, group
        self.deferred_group = deferred_group
        self.poison_halo = poison_halo   # tests: the halo slices of L1..L3 that no exchange fetches hold NaN
        # Rank 0 only, optional (round 5): a second backend (its own context and stream on rank 0's device) for the octaves that are
        # not sharded.  They are gathered on rank 0, and building them is a chain of some eighty small launches -- 0.5 - 0.8 ms of
        # pure latency at 512^3 -- that used to sit between rank 0's slab and its per-keypoint stage, i.e. on the critical path of
        # the whole step.  With a backend of their own they are queued by a second host thread (the library calls release the GIL)
        # and run BESIDE rank 0's per-keypoint stage; their few hundred keypoints are one more run at the end of the record list
        # (csrc/zslab_driver.hip gives them a rank of their own for the same reason).
        self.coarse_be = coarse_backend if (rank == 0 and 0 < plan.n_sharded < len(plan.octaves)) else None
        self._coarse_thread, self._coarse_error = None, None
        self.c_levels, self.c_level_ids = [], []
        self.levels = []       # level table entries, index = level id
        self.level_ids = []
        # exchange_bytes: everything this rank received and sent; of that, deferred_bytes moved in the once-per-octave
        # patch-halo batch that overlaps L4 / L5 / extrema (critical = exchange_bytes - deferred_bytes)
        # hidden_bytes: the part of the per-level halos issued bands-first, i.e. moving while this rank filters its interior
        # per_octave (round 6): [octave] -> bytes moved by that octave's batches and the host time this rank spent completing them
        # (`host_wait_ms`: with gloo that is the transfer as this rank sees it; with nccl = RCCL a completed wait() only orders the
        # rank's stream behind the communicator's, so it is the cost of queueing, and the transfer shows in the step time)
        self.stats = {"exchanges": 0, "exchange_bytes": 0, "deferred_exchanges": 0, "deferred_bytes": 0, "hidden_bytes": 0, "per_octave": {}}
        self._octave_now = 0
        # every DoG buffer an extrema pass was queued on: the library replays those passes from the recorded pointers when
        # a candidate list overflows (cand_finalize), so all five DoG levels of every octave -- not only the L1..L3 / D1..D3
        # the level table names -- must outlive candidates() / describe(); released by the next run()
        self._keepalive = []

    # ---- halo exchange of one level buffer -------------------------------------------------
    def _exchange(self, bufs, z0, z1, e0, width, has_lo, has_hi, inner=0, defer=False, patch=True, spans=None):
        """Refresh, in every buffer of `bufs`, the slices [z0-width, z0-inner) from the lower neighbour and
        [z1+inner, z1+width) from the upper one (and send it the mirror bands of this slab), as ONE batch of
        point-to-point operations.  spans: (inner, width) per buffer instead of one pair for all.  defer=True returns a
        closure that completes the batch (for the caller to run later: the transfers of an RCCL batch proceed on the
        communicator's stream meanwhile); otherwise it is completed here."""
        if not isinstance(bufs, (list, tuple)):
            bufs = [bufs]
        if spans is None:
            spans = [(inner, width)] * len(bufs)
        if self.dist is None or (not has_lo and not has_hi) or all(w <= i for i, w in spans):
            return (lambda: None) if defer else None
        d, ops, back = self.dist, [], []
        grp = self.deferred_group if (defer and patch and self.deferred_group is not None) else self.group
        self.be.before_exchange()
        # gloo cannot move device memory: stage through the host (the single-GPU rehearsals); with
        # nccl (= RCCL) the slices go GPU to GPU over xGMI
        stage = bufs[0].is_cuda and d.get_backend(grp) == "gloo"

        def snd(t):
            return t.cpu() if stage else t

        def rcv(t):
            if not stage:
                return t
            h = t.cpu()
            back.append((t, h))
            return h
        n = 0
        for buf, (inner, width) in zip(bufs, spans):
            if width <= inner:
                continue
            n += width - inner
            if has_lo:   # my slices [z0+inner, z0+width) are the lower neighbour's upper band; its [z0-width, z0-inner) are mine
                ops.append(d.P2POp(d.isend, snd(buf[z0 + inner - e0:z0 + width - e0]), self.rank - 1, grp))
                ops.append(d.P2POp(d.irecv, rcv(buf[z0 - width - e0:z0 - inner - e0]), self.rank - 1, grp))
            if has_hi:
                ops.append(d.P2POp(d.isend, snd(buf[z1 - width - e0:z1 - inner - e0]), self.rank + 1, grp))
                ops.append(d.P2POp(d.irecv, rcv(buf[z1 + inner - e0:z1 + width - e0]), self.rank + 1, grp))
        works = d.batch_isend_irecv(ops)
        nbytes = 2 * n * bufs[0].shape[1] * bufs[0].shape[2] * 4 * (int(has_lo) + int(has_hi))
        self.stats["exchanges"] += 1
        self.stats["exchange_bytes"] += nbytes
        if defer and patch:
            self.stats["deferred_exchanges"] += 1
            self.stats["deferred_bytes"] += nbytes
        elif defer:
            self.stats["hidden_bytes"] += nbytes
        po = self.stats["per_octave"].setdefault(self._octave_now, {"bytes": 0, "critical_bytes": 0, "batches": 0, "host_wait_ms": 0.0})
        po["bytes"] += nbytes
        po["critical_bytes"] += 0 if (defer and patch) else nbytes
        po["batches"] += 1

        def finish():
            import time
            t0 = time.perf_counter()
            for r in works:
                r.wait()
            for t, h in back:
                t.copy_(h)
            self.be.after_exchange()
            po["host_wait_ms"] += 1e3 * (time.perf_counter() - t0)
        if defer:
            return finish
        finish()
        return None

    # ---- one octave on one rank ---------------------------------------------------------------
    def _octave(self, o, L0, z0, z1, e0, e1, zo, has_lo, has_hi, extras, sig, factor, want_next, be=None, sink=None):
        """L0: level-0 buffer covering global slices [e0, e1), exact on slab +- BLUR_HALO (clipped to the volume).
        Returns the L3 buffer (exact on its whole extent) for the subsample that seeds the next octave.
        be / sink: the backend and the (levels, level_ids, keepalive) lists to use (default: the rank's own)."""
        be = self.be if be is None else be
        levels, level_ids, keepalive = (self.levels, self.level_ids, self._keepalive) if sink is None else sink
        nzl = e1 - e0
        shape = (nzl, L0.shape[1], L0.shape[2])
        # As on one device: D0 is read as L0 - L1 around the extrema of D1, and L5 -- hence D4 -- is filtered only around the
        # candidates of D3, from L4: four levels and four halos per octave instead of five, L4's halo nine slices deep (the
        # third extrema phase reads L4 nine slices beyond a candidate).  A backend without that form, or a shape it does
        # not take (rows that are not whole 16-byte vectors), stores every level.
        lazy = hasattr(be, "lazy_ok") and be.lazy_ok(shape, extras[4])
        nlev = 4 if lazy else 5
        L = [L0] + [be.empty(shape) if j <= nlev else None for j in range(1, 6)]
        D = [None if (lazy and j in (0, 4)) else be.empty(shape) for j in range(5)]
        if self.poison_halo:   # nothing writes these slices: a patch that reached them would show in the records
            for l in (1, 2, 3):
                reach = max(PATCH_REACH[l], SUB_HALO if l == 3 else BLUR_HALO)
                if has_lo and z0 - reach > e0:
                    L[l][:z0 - reach - e0] = float("nan")
                if has_hi and e1 > z1 + reach:
                    L[l][z1 + reach - e0:] = float("nan")
        # region recomputed per level: slab +- BLUR_HALO, clipped to what the buffer holds (at a face of the
        # whole volume the buffer ends at the face, which is what makes the zero border exact)
        c0 = max(e0, z0 - BLUR_HALO) if has_lo else e0
        c1 = min(e1, z1 + BLUR_HALO) if has_hi else e1
        a, b = c0 - e0, c1 - e0
        patch_halos = lambda: None
        for j in range(1, nlev + 1):
            hb = BLUR_HALO + (1 if (lazy and j == 4) else 0)
            # Boundary bands first (round 3): what a neighbour fetches of this level are this rank's own first and last hb
            # slices, so where the blur has a windowed form they are filtered first and handed to the exchange, and the
            # interior is filtered while they travel (an RCCL batch runs on the communicator's stream, behind the band
            # launches it was issued after); this rank's own halo slices are not computed, they arrive.  Otherwise: the
            # level on slab +- BLUR_HALO in one piece, then the exchange (round 2).
            banded = (has_lo or has_hi) and hasattr(be, "window_ok") and be.window_ok(shape, extras[j - 1])
            if banded:
                if has_lo:
                    be.blur_dog_window(L[j - 1], L[j], D[j - 1], z0 - e0, z0 - e0 + hb, extras[j - 1])
                if has_hi:
                    be.blur_dog_window(L[j - 1], L[j], D[j - 1], z1 - e0 - hb, z1 - e0, extras[j - 1])
                arrived = self._exchange(L[j], z0, z1, e0, hb, has_lo, has_hi, defer=True, patch=False)
                be.blur_dog_window(L[j - 1], L[j], D[j - 1], (z0 - e0 + hb) if has_lo else 0, (z1 - e0 - hb) if has_hi else nzl,
                                   extras[j - 1])
                arrived()
            else:
                if D[j - 1] is None:
                    be.blur(L[j - 1][a:b], L[j][a:b], extras[j - 1])
                else:
                    be.blur_dog(L[j - 1][a:b], L[j][a:b], D[j - 1][a:b], extras[j - 1])
                # the next blur needs this level exact on slab +- BLUR_HALO: that, and no more, is exchanged here
                self._exchange(L[j], z0, z1, e0, hb, has_lo, has_hi)
            # the fused DoG used the not-yet-exchanged margin of L[j]: redo it on the halo slices
            if has_lo and D[j - 1] is not None:
                be.dog(L[j - 1][a:z0 - e0], L[j][a:z0 - e0], D[j - 1][a:z0 - e0])
            if has_hi and D[j - 1] is not None:
                be.dog(L[j - 1][z1 - e0:b], L[j][z1 - e0:b], D[j - 1][z1 - e0:b])
            if j == 3:
                # L1..L3 are final: what is left of their halos, in two batches that move while L4, L5 and the extrema passes
                # run (round 5; before: one batch of 3 x 24 slices, all of it waited for before the next octave):
                #   the eight slices of L3 beyond +- 8 that the subsample reads -- all the NEXT OCTAVE waits for;
                #   what only patches reach (L1, L2 from 8, L3 from 16, as deep as PATCH_REACH says) -- completed at the end
                #   of run(), i.e. with the whole rest of the pyramid to arrive in.
                patch_halos = self._exchange(L[3], z0, z1, e0, SUB_HALO, has_lo, has_hi, inner=BLUR_HALO, defer=True)
                self._pending.append(self._exchange(L[1:4], z0, z1, e0, HALO, has_lo, has_hi, defer=True,
                                                    spans=[(BLUR_HALO, PATCH_REACH[1]), (BLUR_HALO, PATCH_REACH[2]), (SUB_HALO, PATCH_REACH[3])]))
        for l in range(3):
            lid = o * 3 + l
            if lazy and l == 0:
                be.extrema_append_pair(L[0], L[1], D[1], D[2], lid, z0 - e0, z1 - e0)
            elif lazy and l == 2:
                be.extrema_append_lazy_next(D[2], D[3], L[4], extras[4], lid, z0 - e0, z1 - e0)
            else:
                be.extrema_append(D[l], D[l + 1], D[l + 2], lid, z0 - e0, z1 - e0)
            levels.append(be.level_entry(L[l + 1], D[l + 1], zo, e0, sig[l], sig[l + 1], sig[l + 2], factor))
            level_ids.append(lid)
        patch_halos()   # before the subsample below reads L3 beyond +- BLUR_HALO (the patch-only slices: end of run())
        keepalive.append((D, L))   # a replay of the extrema passes reads D1..D3 and, in the unstored form, L0, L1 and L4
        return L[3] if want_next else None

    # ---- the whole pyramid ----------------------------------------------------------------------
    def run(self, input_slab, input_z0, initial_image_scale=1.0):
        """input_slab: this rank's slices [input_z0, input_z0 + n) of the whole volume, as given by
        SlabPlan.input_range (host array).  Collects the extrema of this rank in the backend."""
        plan, be, rank, S = self.plan, self.be, self.rank, self.plan.nranks
        extra0, extras, sig = sigma_schedule(initial_image_scale)
        K = plan.n_sharded
        self._join_coarse()   # (a run whose results nobody asked for)
        be.reset()
        self.levels, self.level_ids = [], []
        self.c_levels, self.c_level_ids = [], []
        self._keepalive, self._c_keepalive = [], []
        if self.coarse_be is not None:
            self.coarse_be.reset()
        self._pending = []   # the patch-only halo batches of the sharded octaves, completed before run() returns
        try:
            return self._run(input_slab, input_z0, plan, be, rank, S, extra0, extras, sig, K)
        finally:
            for fin in self._pending:   # before anything samples a patch -- and so that no receive outlives its buffer
                fin()
            self._pending = []

    def _run(self, input_slab, input_z0, plan, be, rank, S, extra0, extras, sig, K):
        factor = 1.0
        vol = be.from_host(input_slab)
        nxt = None
        for o in range(len(plan.octaves)):
            X, Y, zo = plan.octaves[o]
            sharded = o < K
            self._octave_now = o
            if sharded:
                z0, z1 = plan.slab(rank, o)
                has_lo, has_hi = rank > 0, rank < S - 1
            else:
                if rank != 0 and K > 0:
                    break               # the gathered octaves live on rank 0
                if K == 0 and rank != 0:
                    break
                z0, z1, has_lo, has_hi = 0, zo, False, False
            e0 = max(0, z0 - HALO) if has_lo else z0
            e1 = min(zo, z1 + HALO) if has_hi else z1
            if o == 0:
                # level 0 = initial blur of the input, computed on slab +- BLUR_HALO from input slab +- 2*BLUR_HALO
                L0 = be.empty((e1 - e0, Y, X))
                i0 = input_z0
                c0 = max(e0, z0 - BLUR_HALO) if has_lo else e0
                c1 = min(e1, z1 + BLUR_HALO) if has_hi else e1
                if i0 >= e0 and hasattr(be, "window_ok") and be.window_ok(tuple(vol.shape), extra0):
                    # round 5: the windowed blur writes planes [c0, c1) straight into the level buffer -- seen from the input's
                    # plane numbering the buffer starts i0 - e0 planes before its own first plane (the input reaches 16 slices
                    # beyond the slab, the buffer 32), and only the window is written: no scratch volume, no copy
                    be.blur_dog_window(vol, L0[i0 - e0:], None, c0 - i0, c1 - i0, extra0)
                else:
                    tmp = be.empty(vol.shape)
                    be.blur(vol, tmp, extra0)
                    L0[c0 - e0:c1 - e0].copy_(tmp[c0 - i0:c1 - i0])
            else:
                L0 = nxt
            want_next = o + 1 < len(plan.octaves)
            L3 = self._octave(o, L0, z0, z1, e0, e1, zo, has_lo, has_hi, extras, sig, factor, want_next)
            factor *= 2.0
            if not want_next:
                break
            Xn, Yn, zn = plan.octaves[o + 1]
            if o + 1 < K:
                # next octave is sharded too: subsample slab +- 2*BLUR_HALO of L3 into next slab +- BLUR_HALO
                n0, n1 = plan.slab(rank, o + 1)
                ne0 = max(0, n0 - HALO) if has_lo else n0
                ne1 = min(zn, n1 + HALO) if has_hi else n1
                s0 = max(ne0, n0 - BLUR_HALO) if has_lo else ne0
                s1 = min(ne1, n1 + BLUR_HALO) if has_hi else ne1
                nxt = be.empty((ne1 - ne0, Yn, Xn))
                be.subsample(L3[2 * s0 - e0:2 * s1 - e0], nxt[s0 - ne0:s1 - ne0])
            elif sharded:
                # last sharded octave: every rank subsamples exactly its slab, rank 0 assembles the whole octave
                t = (z1 - z0) // 2   # an odd last slice of the whole volume is dropped, as in the serial code
                part = be.empty((t, Yn, Xn))
                be.subsample(L3[z0 - e0:z0 - e0 + 2 * t], part)
                nxt = self._gather_to_root(part, zn, Yn, Xn, o)
                if self.coarse_be is not None:
                    # the rest is the other backend's, on a thread of its own: this one goes on to its per-keypoint stage
                    self._start_coarse(o + 1, nxt, factor, extras, sig)
                    break
            else:
                nxt = be.empty((zn, Yn, Xn))
                be.subsample(L3, nxt)
        return self

    def _gather_to_root(self, part, zn, Yn, Xn, o):
        """The per-rank parts of the first gathered octave (the subsampled slabs of octave o) assembled on rank 0 (other ranks get
        None).  Every rank knows every part's size from the plan, so the parts travel as point-to-point transfers straight into their
        slices of the octave: no size exchange (rounds 1 - 4 all_gathered the sizes and read them back -- a host synchronisation on
        every rank in the middle of the pyramid), no padding, no copy out of a gather buffer."""
        d, S = self.dist, self.plan.nranks
        if d is None:
            return part
        torch = __import__("torch")
        places, at = [], 0
        for r in range(S):
            z0, z1 = self.plan.slab(r, o)
            n = max(0, min((z1 - z0) // 2, zn - at))   # an odd last slice of the whole volume is dropped, as in the serial code
            places.append((at, n))
            at += n
        assert at == zn, (at, zn, places)
        stage = part.is_cuda and d.get_backend(self.group) == "gloo"
        self.be.before_exchange()
        if self.rank != 0:
            a, n = places[self.rank]
            if n:
                src = part[:n].cpu() if stage else part[:n]
                for w in d.batch_isend_irecv([d.P2POp(d.isend, src, 0, self.group)]):
                    w.wait()
            self.be.after_exchange()
            return None
        full = self.be.empty((zn, Yn, Xn))
        a, n = places[0]
        full[a:a + n].copy_(part[:n])
        ops, back = [], []
        for r in range(1, S):
            a, n = places[r]
            if not n:
                continue
            dst = full[a:a + n]
            if stage:
                h = torch.empty(dst.shape, dtype=dst.dtype, device="cpu")
                back.append((dst, h))
                dst = h
            ops.append(d.P2POp(d.irecv, dst, r, self.group))
        if ops:
            for w in d.batch_isend_irecv(ops):
                w.wait()
        for t, h in back:
            t.copy_(h)
        self.be.after_exchange()
        return full

    # ---- the octaves that are not sharded, on rank 0's second backend and thread --------------------------------
    def _start_coarse(self, o_first, L0, factor, extras, sig):
        import threading
        torch = self.be.torch
        ready = torch.cuda.Event()
        ready.record(getattr(self.be, "stream", None) or torch.cuda.current_stream())   # the gathered octave is complete behind this
        cbe, plan = self.coarse_be, self.plan
        if hasattr(cbe, "stream"):
            L0.record_stream(cbe.stream)   # allocated on this backend's stream, read on the other's
        sink = (self.c_levels, self.c_level_ids, self._c_keepalive)
        self._c_keepalive.append(L0)

        def work():
            try:
                if hasattr(cbe, "ctx"):
                    torch.cuda.set_device(cbe.ctx.device)   # a new thread starts on device 0, whatever rank 0's device is
                with cbe.stream_scope():
                    (getattr(cbe, "stream", None) or torch.cuda.current_stream()).wait_event(ready)
                    lvl0, f = L0, factor
                    for o in range(o_first, len(plan.octaves)):
                        X, Y, zo = plan.octaves[o]
                        want_next = o + 1 < len(plan.octaves)
                        L3 = self._octave(o, lvl0, 0, zo, 0, zo, zo, False, False, extras, sig, f, want_next, be=cbe, sink=sink)
                        f *= 2.0
                        if not want_next:
                            break
                        Xn, Yn, zn = plan.octaves[o + 1]
                        lvl0 = cbe.empty((zn, Yn, Xn))
                        cbe.subsample(L3, lvl0)
                        self._c_keepalive.append(lvl0)
            except BaseException as e:   # handed to the thread that joins
                self._coarse_error = e
        self._coarse_error = None
        self._coarse_thread = threading.Thread(target=work, name="sift3d-coarse-octaves")
        self._coarse_thread.start()

    def _join_coarse(self):
        t, self._coarse_thread = self._coarse_thread, None
        if t is not None:
            t.join()
        e, self._coarse_error = self._coarse_error, None
        if e is not None:
            raise e

    @staticmethod
    def _table_of(levels, level_ids):
        n = max(level_ids) + 1 if level_ids else 0
        table = [None] * n
        for lid, lv in zip(level_ids, levels):
            table[lid] = lv
        filler = levels[0] if levels else None
        return [t if t is not None else filler for t in table]

    # ---- results ----------------------------------------------------------------------------------
    def _table(self):
        return self._table_of(self.levels, self.level_ids)

    def _coarse_call(self, fn):
        """fn(backend, level table) for the backend of the octaves that are not sharded, on a thread of its own (after that thread's
        pyramid is queued); returns a function that waits for it and returns its result."""
        import threading
        box = {}

        def work():
            try:
                self._join_coarse()
                if hasattr(self.coarse_be, "ctx"):
                    self.coarse_be.torch.cuda.set_device(self.coarse_be.ctx.device)   # (a new thread starts on device 0)
                with self.coarse_be.stream_scope():
                    box["r"] = fn(self.coarse_be, self._table_of(self.c_levels, self.c_level_ids))
            except BaseException as e:
                box["e"] = e
        t = threading.Thread(target=work, name="sift3d-coarse-keypoints")
        t.start()

        def result():
            t.join()
            if "e" in box:
                raise box["e"]
            return box["r"]
        return result

    def candidates(self):
        """This rank's validated extrema (whole-volume coordinates), sorted."""
        if not self.levels:
            return np.zeros(0, dtype=[("octave", "<i4"), ("level", "<i4"), ("is_max", "<i4"), ("x", "<i4"), ("y", "<i4"),
                                      ("z", "<i4"), ("value", "<f4"), ("h_value", "<f4"), ("l_value", "<f4")])
        coarse = self._coarse_call(lambda be, table: be.candidates(table)) if self.coarse_be is not None else None
        mine = self.be.candidates(self._table())
        return mine if coarse is None else np.concatenate([mine, coarse()])   # (the coarser octaves sort behind the sharded ones)

    def describe(self, desc_mode=0, eig_thres=140.0, size_factor=1.0, copy=True):
        """This rank's records and their group ids (level_id*2 + is_max).  copy=False: views of the context's
        download buffers (HIP backend), valid until the next call on the context."""
        if not self.levels:
            return None, np.zeros(0, np.int32)
        if self.coarse_be is not None:   # two contexts, two sets of buffers: one array (the coarser octaves' groups come last)
            coarse = self._coarse_call(lambda be, table: be.describe(table, desc_mode, eig_thres, size_factor, copy=False))
            r0, g0 = self.be.describe(self._table(), desc_mode, eig_thres, size_factor, copy=False)
            r1, g1 = coarse()
            return np.concatenate([r0, r1]), np.concatenate([g0, g1])
        if copy:
            return self.be.describe(self._table(), desc_mode, eig_thres, size_factor)
        return self.be.describe(self._table(), desc_mode, eig_thres, size_factor, copy=False)

    def describe_into(self, shared, desc_mode=0, eig_thres=140.0, size_factor=1.0, group=None, device=None):
        """The per-keypoint stage with every rank's records stored by its own descriptor kernel at their places in `shared` (a
        SharedRecordList): keypoint kernel, the ranks' records per group exchanged (one small all_gather), descriptor kernel, and one
        small all_reduce at the end that tells every rank the record count of the whole volume and that all kernels are done.
        Returns (n, None): shared.view(n) is then the single-GPU list on every rank (rank 0 is the one that uses it; it stays valid
        until a rank's next describe_into).  Or, when the list is too small for the slabs' records, (None, (records, group)): nothing
        was stored in it, this rank's records are views (copies on a rank with two backends) of its context's own buffers as after
        describe(copy=False), and the caller gathers them the old way (gather_records) -- every rank takes the same branch.
        The octaves that are not sharded (rank 0's second backend) take no part in the exchange: their records come last in the
        list whatever the slabs' counts are, so they are described into their own buffers beside everything else and appended by
        rank 0 at the end (a few hundred records)."""
        import torch
        d, be = self.dist, self.be
        groups = shared.pkg.GROUPS
        two = self.coarse_be is not None and bool(self.levels)
        coarse = self._coarse_call(lambda cb, table: cb.describe(table, desc_mode, eig_thres, size_factor, copy=False)) if two else None
        counts = be.describe_counts(self._table(), desc_mode, eig_thres, size_factor)[0] if self.levels else np.zeros(groups, np.int32)
        if d is not None:
            world = d.get_world_size(group)
            dev = "cpu" if d.get_backend(group) == "gloo" else device
            every = [torch.zeros(groups, dtype=torch.int32, device=dev) for _ in range(world)]
            d.all_gather(every, torch.from_numpy(counts).to(dev), group=group)
            allc = np.stack([t.cpu().numpy() for t in every])
            rank = self.rank
        else:
            dev, allc, rank = "cpu", counts[None, :], 0
        shift, slab_total = placed_shifts(allc, rank)
        fits = slab_total <= shared.capacity
        own = (None, np.zeros(0, np.int32))
        if self.levels:
            res = be.describe_place(shared.address if fits else None, shift if fits else None)
            if not fits:
                own = res
        crecs, cgrp = coarse() if two else (None, None)
        if not fits:
            if two:
                own = (np.concatenate([own[0], crecs]), np.concatenate([own[1], cgrp]))
            return None, own
        total = slab_total
        shared.overflow = None
        if two and len(crecs):
            total = slab_total + len(crecs)
            if total <= shared.capacity:
                shared.records[slab_total:total] = crecs
            else:   # no room behind the slabs' records (a list sized too tightly): rank 0 keeps a list of its own for this run
                shared.overflow = np.concatenate([shared.records[:slab_total], crecs])
        if d is not None:   # every rank's kernel has stored its records (describe_place ends with a stream synchronisation); rank 0 knows the total
            t = torch.tensor([total if rank == 0 else 0], dtype=torch.int64, device=dev)
            d.all_reduce(t, op=d.ReduceOp.MAX, group=group)
            total = int(t.item())
        return total, None


class SharedRecordList:
    """ONE list of records in shared memory that every rank's process maps and registers with its device: the ranks' descriptor
    kernels store their records straight into their places of the single-GPU order (ZSlabExtractor.describe_into), so nothing is
    gathered and nothing is merged -- what csrc/zslab_driver.hip does inside one process, across processes.  (gather_records sends
    every rank's records to rank 0 through the collective backend: with RCCL that is an upload of the records each rank has just
    downloaded, the gather, and a download of all of them on rank 0 -- 63 MB at 512^3, behind the last kernel.)
    Collective: every rank of `dist` calls it with the same capacity."""

    _count = 0

    def __init__(self, pkg, dist, rank, capacity, dtype, group=None):
        import ctypes
        import mmap
        import os
        self.pkg, self.capacity, self.rank = pkg, int(capacity), rank
        self.nbytes = max(1, self.capacity) * np.dtype(dtype).itemsize
        self._mm, self._cbuf, self.records, self.overflow, self.address = None, None, None, None, 0
        SharedRecordList._count += 1
        # Every collective below is reached by every rank whatever went wrong locally: a rank that cannot create, map or register the
        # segment says so in the last one, and then EVERY rank raises -- nobody is left waiting in a broadcast for a rank that gave up.
        err, name, fd = None, [None], -1
        if rank == 0:
            try:
                name = ["/dev/shm/sift3d_records_%d_%d" % (os.getpid(), SharedRecordList._count)]
                fd = os.open(name[0], os.O_RDWR | os.O_CREAT | os.O_EXCL, 0o600)
                os.ftruncate(fd, self.nbytes)
            except Exception as e:
                err, name = e, [None]
        if dist is not None:
            dist.broadcast_object_list(name, src=0, group=group)
        try:
            if name[0] is None:
                raise err or RuntimeError("rank 0 could not create the shared segment")
            if rank != 0:
                fd = os.open(name[0], os.O_RDWR)
            self._mm = mmap.mmap(fd, self.nbytes)
        except Exception as e:
            err = err or e
        if fd >= 0:
            os.close(fd)
        if dist is not None:
            dist.barrier(group=group)   # every rank has it mapped (or has failed): the name can go -- the pages live as long as a mapping does
        if rank == 0 and name[0] is not None:
            try:
                os.unlink(name[0])
            except OSError:
                pass
        if err is None:
            try:
                self._cbuf = (ctypes.c_char * self.nbytes).from_buffer(self._mm)
                self.address = ctypes.addressof(self._cbuf)
                pkg.host_register(self.address, self.nbytes)
                self.records = np.frombuffer(self._cbuf, dtype, self.capacity)
            except Exception as e:
                err, self.address = e, 0
        oks = [err is None]
        if dist is not None:
            oks = [None] * dist.get_world_size(group)
            dist.all_gather_object(oks, err is None, group=group)
        if not all(oks):
            self.close()
            raise RuntimeError("no shared record list: %s" % (err if err is not None else "rank(s) %s could not set it up" % [i for i, k in enumerate(oks) if not k]))
        # self.overflow: rank 0, after a describe_into whose coarse octaves' records did not fit behind the slabs': the whole list

    def view(self, n):
        """The n records of the last describe_into."""
        return self.overflow if self.overflow is not None else self.records[:n]

    def close(self):
        if self._mm is None:
            return
        if self.address:
            self.pkg.host_unregister(self.address)
            self.address = 0
        self.records = None
        self._cbuf = None
        try:
            self._mm.close()
        except BufferError:   # a view of the list is still referenced somewhere: the mapping goes with the last of them
            pass
        self._mm = None


def placed_shifts(all_counts, rank):
    """all_counts[r][g]: rank r's records of group g.  The single-GPU order is, group by group, one run per rank in rank order
    (within a group slabs are in z order); a rank's own records are sorted by group already, so its record i of group g goes to
    i + shift[g].  Returns (shift of `rank`, total)."""
    c = np.asarray(all_counts, np.int64)                      # (world, GROUPS)
    before_groups = np.concatenate([[0], np.cumsum(c.sum(axis=0))[:-1]])          # records of all ranks in groups < g
    before_ranks = np.cumsum(c, axis=0)[rank] - c[rank]                            # records of lower ranks in group g
    own_before = np.concatenate([[0], np.cumsum(c[rank])[:-1]])                    # this rank's own records in groups < g
    return (before_groups + before_ranks - own_before).astype(np.int32), int(c.sum())


def merge_by_group(parts):
    """parts: per rank (records, group) in rank order -> records in the single-GPU order.
    Within a group (level, is_max) slabs are in z order, so rank order is raster order."""
    recs = [p[0] for p in parts if p[0] is not None and len(p[0])]
    grps = [p[1] for p in parts if p[0] is not None and len(p[0])]
    if not recs:
        return None
    allr, allg = np.concatenate(recs), np.concatenate(grps)
    order = np.argsort(allg, kind="stable")   # stable: keeps rank order, and raster order inside a rank
    return allr[order]


def gather_records(dist, rank, world, recs, grp, device, group=None, dtype=None):
    """Gather every rank's (records, group) on rank 0 with tensor collectives (sizes first, then padded
    byte tensors) and merge them into the single-GPU order.  Returns the merged array on rank 0, None elsewhere.
    dtype: the record dtype (the package's FEATURE_DTYPE); needed when rank 0 itself has no records."""
    import torch
    stage = dist.get_backend(group) == "gloo"
    dev = "cpu" if stage else device
    n = 0 if recs is None else len(recs)
    item = 0 if recs is None else recs.dtype.itemsize
    sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n, item], dtype=torch.int64, device=dev), group=group)
    counts = [int(t[0].item()) for t in sizes]
    item = max(int(t[1].item()) for t in sizes)
    mx = max(counts)
    if mx == 0:
        return None
    rb = torch.zeros(mx * item, dtype=torch.uint8, device=dev)
    gb = torch.zeros(mx, dtype=torch.int32, device=dev)
    if n:
        rb[:n * item].copy_(torch.from_numpy(recs.view(np.uint8).reshape(-1)))
        gb[:n].copy_(torch.from_numpy(grp.astype(np.int32)))
    if rank == 0:
        rbs = [torch.zeros(mx * item, dtype=torch.uint8, device=dev) for _ in range(world)]
        gbs = [torch.zeros(mx, dtype=torch.int32, device=dev) for _ in range(world)]
        dist.gather(rb, rbs, dst=0, group=group)
        dist.gather(gb, gbs, dst=0, group=group)
        dt = np.dtype(dtype) if dtype is not None else (recs.dtype if recs is not None else None)
        if dt is None:
            raise ValueError("gather_records: rank 0 has no records of its own, pass dtype=")
        assert dt.itemsize == item, (dt.itemsize, item)
        parts = []
        for r in range(world):
            if counts[r] == 0:
                continue
            a = rbs[r][:counts[r] * item].cpu().numpy()
            parts.append((a.view(dt), gbs[r][:counts[r]].cpu().numpy()))
        return merge_by_group(parts)
    dist.gather(rb, None, dst=0, group=group)
    dist.gather(gb, None, dst=0, group=group)
    return None
