"""GPU suite: the HIP path through the C-ABI against the oracle.

Bar: bit-exact for every integer / index output and, because the kernels
repeat the reference's float operations in the same order without FMA
contraction, bit-exact for the float volumes too; the pipeline's float record
fields are additionally held to the 1e-4 tolerance the north star states.
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json north_star: descriptor / geometry floats within 1e-4


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def vol_of(built, dims, seed):
    return built.synth_blobs(*dims, seed=seed)


SHAPES = [(17, 13, 11), (32, 32, 32), (64, 48, 40), (33, 21, 19), (130, 6, 9), (5, 70, 7), (256, 8, 8), (300, 20, 12)]
SIGMAS = [0.5, 0.95, 1.2262736558914185, 1.5198684930801392, 1.5450079441070557, 1.9465880393981934,
          2.452547311782837, 3.0900158882141113]


@pytest.mark.parametrize("dims", SHAPES)
def test_blur_bit_exact(built, oracle, dims):
    vol = vol_of(built, dims, 11) + np.float32(3.0)
    with built.Context(*dims) as ctx:
        for s in SIGMAS:
            got = ctx.gauss_blur(vol, s)
            want = oracle.blur(vol, s)
            assert (bits(got) == bits(want)).all(), (dims, s, np.abs(got - want).max())


FUSED_SHAPES = [(32, 32, 32), (64, 48, 40), (256, 8, 8), (300, 20, 12), (132, 37, 45), (68, 17, 3), (128, 64, 70), (260, 50, 21)]


@pytest.mark.parametrize("dims", FUSED_SHAPES)
@pytest.mark.parametrize("chunks,rows,tile,stagger", [(0, 0, 0, 0), (3, 1, 0, 0), (2, 2, 1, 0), (3, 2, 1, 0), (1, 1, 0, 0), (5, 0, 0, 0), (0, 2, 2, 0), (3, 2, 2, 0),
                                                      (0, 2, 1, 1), (3, 2, 2, 1), (0, 2, 1, 2), (3, 2, 2, 2), (2, 2, 0, 2), (1, 2, 2, 1)])
def test_fused_blur_dog_bit_exact(built, oracle, dims, chunks, rows, tile, stagger):
    """The one-launch x+y+z+DoG kernel (forced on: the pipeline only uses it from 2^22 voxels up, 2^18 for narrow filters):
    partial tiles in x and y, volumes thinner than the filter, one or several z chunks, both thread mappings (one row per
    thread with one plane of window prefetch; two rows with two planes) on every shape and filter: level and DoG
    bit-identical to the oracle, for level + DoG, level only and DoG only.  stagger (round 6, SIFT3D_TUNE_FUSED_STAGGER): 0 the
    default (on from 11 taps up), 1 off (the kernel of rounds 2 - 5), 2 on for every filter it is built for (7 - 13 taps): one copy
    of the march per wavefront role and the second half of the wavefronts half a step behind the first."""
    import torch
    vol = vol_of(built, dims, 5) - np.float32(1.5)
    nx, ny, nz = dims
    with built.Context(*dims) as ctx:
        ctx.set_tuning(built.TUNE_BLUR_FUSED, 2)
        ctx.set_tuning(built.TUNE_FUSED_CHUNKS, chunks)
        ctx.set_tuning(built.TUNE_FUSED_ROWS, rows)
        ctx.set_tuning(built.TUNE_FUSED_TILE, tile)   # 2: the 128 x 16 tile (rows of at least 128 voxels; round 4)
        ctx.set_tuning(built.TUNE_FUSED_STAGGER, stagger)
        d_in = torch.from_numpy(vol).cuda()
        d_out, d_dog = torch.empty_like(d_in), torch.empty_like(d_in)
        torch.cuda.synchronize()
        ctx.enable_timing(True)
        for s in SIGMAS[1:]:
            want = oracle.blur(vol, s)
            ctx.gauss_blur_dog_dev(d_in.data_ptr(), d_out.data_ptr(), d_dog.data_ptr(), nx, ny, nz, s)
            ctx.sync()
            assert (bits(d_out.cpu().numpy()) == bits(want)).all(), (dims, s)
            assert (bits(d_dog.cpu().numpy()) == bits(oracle.dog(vol, want))).all(), (dims, s)
            d_out.zero_()
            torch.cuda.synchronize()   # torch's stream and the context's stream are not ordered with each other
            ctx.gauss_blur_dev(d_in.data_ptr(), d_out.data_ptr(), nx, ny, nz, s)   # no DoG output
            ctx.sync()
            assert (bits(d_out.cpu().numpy()) == bits(want)).all(), (dims, s)
            d_dog.zero_()
            torch.cuda.synchronize()
            ctx.gauss_blur_dog_dev(d_in.data_ptr(), 0, d_dog.data_ptr(), nx, ny, nz, s)   # DoG only (the sixth level)
            ctx.sync()
            assert (bits(d_dog.cpu().numpy()) == bits(oracle.dog(vol, want))).all(), (dims, s)
        log = ctx.launch_log()
        assert (log["stage"] == built.STAGES.index("blur_fused")).all()   # the fused kernel is what ran


@pytest.mark.parametrize("dims", [(64, 40, 30), (128, 64, 33), (256, 48, 20), (512, 64, 16), (1024, 32, 12), (192, 40, 17), (320, 33, 9)])
@pytest.mark.parametrize("order", [1, 2, 3])
def test_fused_blur_tile_orders_cover_every_tile_once(built, oracle, dims, order):
    """Round 5: which workgroup takes which tile of the fused launch is a knob (SIFT3D_TUNE_FUSED_ORDER: rows of tiles per XCD,
    workgroup b = tile b, column strips per XCD).  Every order must hand out every (tile, chunk) exactly once: level and DoG
    bit-identical to the oracle for 1, 2, 4, 8 and 16 tile columns (where the strip order applies: the counts divide), 3 and 5
    columns (where the launcher falls back), both tiles, one and several z chunks (also counts the strips cannot split evenly)."""
    import torch
    vol = vol_of(built, dims, 23) + np.float32(0.5)
    nx, ny, nz = dims
    want = {s: oracle.blur(vol, s) for s in (SIGMAS[2], SIGMAS[5])}
    with built.Context(*dims) as ctx:
        ctx.set_tuning(built.TUNE_BLUR_FUSED, 2)
        ctx.set_tuning(built.TUNE_FUSED_ORDER, order)
        d_in = torch.from_numpy(vol).cuda()
        d_out, d_dog = torch.empty_like(d_in), torch.empty_like(d_in)
        torch.cuda.synchronize()
        for tile, rows, chunks in ((1, 2, 0), (2, 2, 0), (1, 2, 3), (2, 2, 2), (1, 1, 0), (2, 2, 5)):
            ctx.set_tuning(built.TUNE_FUSED_TILE, tile)
            ctx.set_tuning(built.TUNE_FUSED_ROWS, rows)
            ctx.set_tuning(built.TUNE_FUSED_CHUNKS, chunks)
            for s, w in want.items():
                d_out.fill_(-7.0); d_dog.fill_(-7.0)
                torch.cuda.synchronize()
                ctx.gauss_blur_dog_dev(d_in.data_ptr(), d_out.data_ptr(), d_dog.data_ptr(), nx, ny, nz, s)
                ctx.sync()
                assert (bits(d_out.cpu().numpy()) == bits(w)).all(), (dims, order, tile, rows, chunks, s)
                assert (bits(d_dog.cpu().numpy()) == bits(oracle.dog(vol, w))).all(), (dims, order, tile, rows, chunks, s)


@pytest.mark.parametrize("dims", [(64, 48, 40), (136, 37, 45), (72, 17, 3), (128, 64, 70), (264, 50, 21), (16, 2, 2), (40, 33, 12)])
@pytest.mark.parametrize("chunks,tile,stagger", [(0, 0, 0), (1, 1, 0), (3, 1, 0), (5, 2, 0), (2, 2, 0), (0, 0, 1), (3, 1, 2), (5, 2, 1), (2, 2, 2)])
def test_fused_blur_carries_the_half_size_volume(built, oracle, dims, chunks, tile, stagger):
    """Round 4: the launch that makes level 3 (11 taps, two rows per thread) also writes the next octave's level 0, the
    2 x 2 x 2 mean of the level, from the planes it holds in registers.  Level, DoG and the half-size volume against the oracle's
    blur -> subsample, with odd ny / nz (the last row / plane has no partner), one and several z chunks (a pair of planes must not
    straddle two), both tiles; and the same bytes from the two-launch form (TUNE_FUSED_SUB 0) and for a filter the carry is
    not built for."""
    import torch
    vol = vol_of(built, dims, 5) - np.float32(1.5)
    nx, ny, nz = dims
    s11 = [s for s in SIGMAS if len(oracle.taps(s)) == 11][0]
    s_other = [s for s in SIGMAS if len(oracle.taps(s)) == 9][0]
    with built.Context(*dims) as ctx:
        ctx.set_tuning(built.TUNE_BLUR_FUSED, 2)
        ctx.set_tuning(built.TUNE_FUSED_CHUNKS, chunks)
        ctx.set_tuning(built.TUNE_FUSED_ROWS, 2)
        ctx.set_tuning(built.TUNE_FUSED_TILE, tile)
        ctx.set_tuning(built.TUNE_FUSED_STAGGER, stagger)
        d_in = torch.from_numpy(vol).cuda()
        d_out, d_dog = torch.empty_like(d_in), torch.empty_like(d_in)
        d_half = torch.full((nz // 2, ny // 2, nx // 2), 7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.enable_timing(True)
        for sigma, sub, expect_one in ((s11, 1, True), (s11, 0, False), (s_other, 1, False)):
            want = oracle.blur(vol, sigma)
            ctx.set_tuning(built.TUNE_FUSED_SUB, sub)
            d_half.fill_(7.0)
            d_out.zero_()
            torch.cuda.synchronize()
            one = ctx.gauss_blur_dog_half_dev(d_in.data_ptr(), d_out.data_ptr(), d_dog.data_ptr(), d_half.data_ptr(), nx, ny, nz, sigma)
            ctx.sync()
            assert one == expect_one, (dims, sigma, sub)
            assert (bits(d_out.cpu().numpy()) == bits(want)).all(), (dims, sigma)
            assert (bits(d_dog.cpu().numpy()) == bits(oracle.dog(vol, want))).all(), (dims, sigma)
            assert (bits(d_half.cpu().numpy()) == bits(oracle.subsample(want))).all(), (dims, sigma, sub)
        log = ctx.launch_log()
        names = [built.STAGES[i] for i in log["stage"]]
        assert names == ["blur_fused", "blur_fused", "subsample", "blur_fused", "subsample"]


def test_blur_dog_half_refuses_what_it_cannot_do(built):
    """sift3d_gauss_blur_dog_half_dev: null pointers, a dimension below 2 (nothing to halve) and a volume beyond the context are
    errors with a message, not launches."""
    import torch
    with built.Context(32, 32, 32) as ctx:
        a = torch.zeros(32, 32, 32, device="cuda")
        o, d, h = torch.empty_like(a), torch.empty_like(a), torch.empty(16, 16, 16, device="cuda")
        torch.cuda.synchronize()
        for args in ((0, o.data_ptr(), d.data_ptr(), h.data_ptr(), 32, 32, 32), (a.data_ptr(), 0, d.data_ptr(), h.data_ptr(), 32, 32, 32),
                     (a.data_ptr(), o.data_ptr(), d.data_ptr(), 0, 32, 32, 32), (a.data_ptr(), o.data_ptr(), d.data_ptr(), h.data_ptr(), 32, 32, 1),
                     (a.data_ptr(), o.data_ptr(), d.data_ptr(), h.data_ptr(), 1, 32, 32), (a.data_ptr(), o.data_ptr(), d.data_ptr(), h.data_ptr(), 64, 64, 64)):
            with pytest.raises(RuntimeError):
                ctx.gauss_blur_dog_half_dev(*args, 1.9465880393981934)
        # and the DoG output is optional
        assert ctx.gauss_blur_dog_half_dev(a.data_ptr(), o.data_ptr(), 0, h.data_ptr(), 32, 32, 32, 1.9465880393981934) is False
        ctx.sync()
        assert float(h.abs().max()) == 0.0


@pytest.mark.parametrize("dims", [(64, 48, 40), (132, 70, 33), (256, 8, 24)])
def test_windowed_blur_writes_exactly_its_planes(built, oracle, dims):
    """sift3d_gauss_blur_dog_window_dev (what a Z-slab rank filters its boundary bands with): the planes of the window are the
    full blur's planes bit for bit -- level and DoG -- for windows at the faces, in the middle, of one plane and of the whole
    volume, with one and several z chunks and both thread mappings; every plane outside the window keeps what it held."""
    import torch
    nx, ny, nz = dims
    vol = vol_of(built, dims, 9) - np.float32(2.0)
    with built.Context(*dims) as ctx:
        d_in = torch.from_numpy(vol).cuda()
        torch.cuda.synchronize()
        for sigma in (SIGMAS[2], SIGMAS[4], SIGMAS[6], SIGMAS[7]):        # 7, 9, 13, 17 taps
            assert ctx.blur_window_supported(nx, ny, sigma)
            want = oracle.blur(vol, sigma)
            want_dog = oracle.dog(vol, want)
            for (lo, hi), chunks, rows in (((0, 5), 0, 0), ((nz - 3, nz), 2, 1), ((7, 8), 0, 2), ((4, nz - 6), 3, 0), ((0, nz), 0, 0)):
                ctx.set_tuning(built.TUNE_FUSED_CHUNKS, chunks)
                ctx.set_tuning(built.TUNE_FUSED_ROWS, rows)
                d_out = torch.full_like(d_in, 123.0)
                d_dog = torch.full_like(d_in, -321.0)
                torch.cuda.synchronize()
                ctx.gauss_blur_dog_window_dev(d_in.data_ptr(), d_out.data_ptr(), d_dog.data_ptr(), nx, ny, nz, lo, hi, sigma)
                ctx.sync()
                o, g = d_out.cpu().numpy(), d_dog.cpu().numpy()
                assert (bits(o[lo:hi]) == bits(want[lo:hi])).all() and (bits(g[lo:hi]) == bits(want_dog[lo:hi])).all(), (sigma, lo, hi)
                assert (o[:lo] == 123.0).all() and (o[hi:] == 123.0).all() and (g[:lo] == -321.0).all() and (g[hi:] == -321.0).all()
        assert not ctx.blur_window_supported(nx + 1, ny, SIGMAS[2])        # rows that are not whole 16-byte vectors
        with pytest.raises(built.Sift3DError):
            ctx.gauss_blur_dog_window_dev(d_in.data_ptr(), d_in.data_ptr(), 0, nx, ny, nz, 3, 3, SIGMAS[2])
        # the slice accessor of the command line's image.pgm
        ctx.set_volume(vol)
        ctx.detect()
        sl = np.empty((ny, nx), np.float32)
        from ctypes import c_void_p
        rc = built.hip_lib().sift3d_get_level_slice(c_void_p(ctx.handle), 0, 0, nz // 2, c_void_p(sl.ctypes.data), None, None)
        assert rc == 0 and (bits(sl) == bits(oracle.blur(vol, SIGMAS[3])[nz // 2])).all()


def test_dev_entry_points_are_ordered_with_the_default_stream(built, oracle):
    """A caller on the default stream (torch's, and the reference's own) needs no synchronisation around the *_dev entry
    points: the input is still being produced by queued torch kernels when sift3d_gauss_blur_dog_dev is called, and torch
    consumes the outputs right after the call returns.  Repeated with the buffers rewritten each round, so that a missing
    fence shows as stale or half-written data."""
    import torch
    dims = (96, 80, 72)
    nx, ny, nz = dims
    sigma = 1.9465880393981934
    base = vol_of(built, dims, 8)
    want = {}
    for k in range(4):
        v = (base * np.float32(1 + k) + np.float32(k)).astype(np.float32)
        b = oracle.blur(v, sigma)
        want[k] = (b, oracle.dog(v, b), oracle.subsample(b))
    with built.Context(*dims) as ctx:
        d_base = torch.from_numpy(base).cuda()
        d_in = torch.empty_like(d_base)
        d_out, d_dog = torch.empty_like(d_base), torch.empty_like(d_base)
        d_half = torch.empty((nz // 2, ny // 2, nx // 2), dtype=torch.float32, device="cuda")
        junk = torch.empty((64, 1024, 1024), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for rnd in range(3):
            for k in range(4):
                for _ in range(3):
                    junk.normal_()                                        # keeps the default stream busy ahead of the producer
                d_in.copy_(d_base * float(1 + k) + float(k))              # producer: queued, not finished
                d_out.fill_(float("nan")); d_dog.fill_(float("nan")); d_half.fill_(float("nan"))
                ctx.gauss_blur_dog_dev(d_in.data_ptr(), d_out.data_ptr(), d_dog.data_ptr(), nx, ny, nz, sigma)
                ctx.subsample2_dev(d_out.data_ptr(), nx, ny, nz, d_half.data_ptr())
                got = (d_out.clone(), d_dog.clone(), d_half.clone())      # consumers on the default stream, no ctx.sync()
                d_in.fill_(-1.0)                                          # and the input is overwritten straight away
                for g, w, name in zip(got, want[k], ("level", "dog", "half")):
                    assert (bits(g.cpu().numpy()) == bits(w)).all(), (rnd, k, name)
    # the sequence that failed once in round 1 (before the fences): a tiny volume, the output cleared on the default
    # stream immediately before the call.  The race is timing-dependent and could not be provoked on demand on the test
    # boxes even without the fences (tools/fence_probe.py: 0 of 200), so this is a regression case, not a proof.
    sd = (256, 8, 8)
    sv = vol_of(built, sd, 5) - np.float32(1.5)
    sw = oracle.blur(sv, 1.5198684930801392)
    with built.Context(*sd) as ctx:
        d_in = torch.from_numpy(sv).cuda()
        d_out = torch.empty_like(d_in)
        for it in range(50):
            d_out.zero_()
            ctx.gauss_blur_dev(d_in.data_ptr(), d_out.data_ptr(), sd[0], sd[1], sd[2], 1.5198684930801392)
            assert (bits(d_out.clone().cpu().numpy()) == bits(sw)).all(), it


def test_slab_entry_points_are_ordered_with_the_default_stream(built):
    """The same for the slab building blocks (round-2 advice): levels produced by queued work on the default stream,
    sift3d_extrema_append_dev / _append_lazy_dev, then sift3d_describe_dev, with no synchronisation anywhere; the DoG levels
    only the extrema passes read are overwritten on the default stream right after the appends (fence out), the level table's
    buffers stay.  Expected: the first octave of the single-context extraction, byte for byte."""
    import importlib
    import torch
    zs = importlib.import_module("3d_sift_cuda_amd.zslab")
    dims = (96, 88, 80)
    nx, ny, nz = dims
    vol = vol_of(built, dims, 21)
    extra0, extras, sig = zs.sigma_schedule(1.0)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        ctx.set_max_octaves(1)
        want = ctx.extract()
    assert len(want) > 100
    for lazy in (False, True):
        with built.Context(nx, ny, nz, slab=True) as ctx:
            d_vol = torch.from_numpy(vol).cuda()
            junk = torch.empty((64, 1024, 1024), dtype=torch.float32, device="cuda")
            L = [torch.empty_like(d_vol) for _ in range(6)]
            D = [torch.empty_like(d_vol) for _ in range(5)]
            torch.cuda.synchronize()
            for rnd in range(3):
                for t in L + D:
                    t.fill_(float("nan"))
                for _ in range(3):
                    junk.normal_()                                      # the default stream is busy ahead of everything
                src = d_vol * 1.0                                       # the input itself is still being produced
                ctx.gauss_blur_dev(src.data_ptr(), L[0].data_ptr(), nx, ny, nz, extra0)
                for j in range(1, 6):
                    ctx.gauss_blur_dog_dev(L[j - 1].data_ptr(), L[j].data_ptr(), D[j - 1].data_ptr(), nx, ny, nz, extras[j - 1])
                ctx.candidates_reset()
                if lazy:
                    assert ctx.lazy_levels_supported(nx, ny, nz, extras[4])
                    ctx.extrema_append_lazy_dev(0, L[0].data_ptr(), L[1].data_ptr(), D[1].data_ptr(), D[2].data_ptr(), 0, 0.0,
                                                nx, ny, nz, 0, 0, nz)
                    ctx.extrema_append_dev(D[1].data_ptr(), D[2].data_ptr(), D[3].data_ptr(), nx, ny, nz, 1, 0, nz)
                    ctx.extrema_append_lazy_dev(D[2].data_ptr(), 0, 0, D[3].data_ptr(), 0, L[4].data_ptr(), extras[4],
                                                nx, ny, nz, 2, 0, nz)
                else:
                    for l in range(3):
                        ctx.extrema_append_dev(D[l].data_ptr(), D[l + 1].data_ptr(), D[l + 2].data_ptr(), nx, ny, nz, l, 0, nz)
                D[0].fill_(-7.0); D[4].fill_(7.0); L[5].fill_(0.0)        # nothing after the appends reads them
                if not lazy:
                    L[0].fill_(3.0)
                levels = [{"img": L[l + 1].data_ptr(), "dogc": D[l + 1].data_ptr(), "nx": nx, "ny": ny, "nz_local": nz,
                           "nz_global": nz, "z_offset": 0, "sigma_h": sig[l], "sigma_c": sig[l + 1], "sigma_l": sig[l + 2],
                           "octave_factor": 1.0} for l in range(3)]
                got, _ = ctx.describe_dev(levels)
                assert got.tobytes() == want.tobytes(), (lazy, rnd)


def test_describe_in_two_parts_places_the_records_where_the_shifts_say(built):
    """sift3d_describe_dev_counts / sift3d_describe_dev_place (round 5): the per-keypoint stage of a context whose records go into a list
    it shares with others.  One context here, and an imaginary rank in front of it whose records per group are made up: this context's
    records of group g must land behind that rank's run of group g, in their own order, and the group counts must be the histogram of
    the group words.  Also: the list too small (place with no list: the context's own buffers, as describe_dev), and place without
    counts refused."""
    import importlib
    import torch
    zs = importlib.import_module("3d_sift_cuda_amd.zslab")
    dims = (96, 88, 80)
    nx, ny, nz = dims
    vol = vol_of(built, dims, 33)
    extra0, extras, sig = zs.sigma_schedule(1.0)
    with built.Context(nx, ny, nz, slab=True) as ctx:
        d_vol = torch.from_numpy(vol).cuda()
        L = [torch.empty_like(d_vol) for _ in range(6)]
        D = [torch.empty_like(d_vol) for _ in range(5)]
        levels = [{"img": L[l + 1].data_ptr(), "dogc": D[l + 1].data_ptr(), "nx": nx, "ny": ny, "nz_local": nz, "nz_global": nz, "z_offset": 0,
                   "sigma_h": sig[l], "sigma_c": sig[l + 1], "sigma_l": sig[l + 2], "octave_factor": 1.0} for l in range(3)]

        def pyramid():
            ctx.gauss_blur_dev(d_vol.data_ptr(), L[0].data_ptr(), nx, ny, nz, extra0)
            for j in range(1, 6):
                ctx.gauss_blur_dog_dev(L[j - 1].data_ptr(), L[j].data_ptr(), D[j - 1].data_ptr(), nx, ny, nz, extras[j - 1])
            ctx.candidates_reset()
            for l in range(3):
                ctx.extrema_append_dev(D[l].data_ptr(), D[l + 1].data_ptr(), D[l + 2].data_ptr(), nx, ny, nz, l, 0, nz)
        pyramid()
        want, wgrp = ctx.describe_dev(levels)
        assert len(want) > 100
        with pytest.raises(built.Sift3DError):
            ctx.describe_dev_place(None, None)                                # no counts before it
        pyramid()
        counts, n = ctx.describe_dev_counts(levels)
        assert n == len(want) and (counts == np.bincount(wgrp, minlength=built.GROUPS)).all()
        rng = np.random.default_rng(1)
        other = np.where(counts > 0, rng.integers(0, 7, built.GROUPS), 0).astype(np.int32)   # the imaginary rank 0 in front
        shift0, total = zs.placed_shifts(np.stack([other, counts]), 0)
        shift1, _ = zs.placed_shifts(np.stack([other, counts]), 1)
        assert total == n + int(other.sum())
        lst = np.zeros(total + 5, built.FEATURE_DTYPE)
        lst["scale"] = -1.0                                                    # what nobody stores stays as it is
        built.host_register(lst.ctypes.data, lst.nbytes)
        try:
            assert ctx.describe_dev_place(lst.ctypes.data, shift1) == n
            at, mine = 0, np.zeros(total + 5, bool)
            pos = 0
            for g in range(built.GROUPS):
                pos += int(other[g])                                           # the other rank's run of the group comes first
                k = int(counts[g])
                assert lst[pos:pos + k].tobytes() == want[at:at + k].tobytes(), g
                mine[pos:pos + k] = True
                pos += k; at += k
            assert (lst["scale"][~mine] == -1.0).all()
            # the list too small: every rank would fall back to its own buffers -- place with no list
            pyramid()
            ctx.describe_dev_counts(levels)
            own, ogrp = ctx.describe_dev_place(None, None)
            assert own.tobytes() == want.tobytes() and (ogrp == wgrp).all()
        finally:
            built.host_unregister(lst.ctypes.data)


def test_lds_float_atomic_add_rounds_like_the_alu(built):
    """The orientation-histogram splat adds with ds_add_f32: it must be the IEEE add the reference's CPU performs."""
    rng = np.random.default_rng(7)
    n = 1 << 16
    a = (rng.standard_normal(n) * 10.0 ** rng.integers(-42, 30, n)).astype(np.float32)
    b = (rng.standard_normal(n) * 10.0 ** rng.integers(-42, 30, n)).astype(np.float32)
    b[:n // 4] = -a[:n // 4] * np.float32(1.0000001)                      # cancellation
    a[n // 4:n // 2] = a[n // 4:n // 2] * np.float32(1e-39) / np.float32(1e-3)  # denormal operands and results
    b[n // 4:n // 2] = b[n // 4:n // 2] * np.float32(1e-39) / np.float32(1e-3)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, 1.0, -1.0, 0.5 ** 24],
                       np.float32)
    aa, bb = np.meshgrid(special, special)
    a[-aa.size:], b[-bb.size:] = aa.ravel(), bb.ravel()
    with built.Context(64, 64, 64) as ctx:
        valu, lds = ctx.selftest_lds_add(a, b)
    with np.errstate(all="ignore"):
        want = a + b
    same = lambda p, q: ((bits(p) == bits(q)) | (np.isnan(p) & np.isnan(q))).all()
    assert same(valu, want) and same(lds, want)
    fin = ~np.isnan(want)
    assert (bits(lds[fin]) == bits(valu[fin])).all()


def test_blur_generic_tap_counts(built, oracle):
    """sigmas outside the templated 3..17-tap range take the generic kernel."""
    dims = (40, 24, 20)
    vol = vol_of(built, dims, 2)
    with built.Context(*dims) as ctx:
        for s in (0.0, 0.2, 4.5, 7.0):
            assert (bits(ctx.gauss_blur(vol, s)) == bits(oracle.blur(vol, s))).all(), s


def test_blur_random_noise_and_extremes(built, oracle):
    rng = np.random.default_rng(0)
    dims = (48, 36, 28)
    vol = (rng.standard_normal(dims[::-1]) * 1000).astype(np.float32)
    vol[0, 0, :8] = [0, -0.0, 1e-38, -1e-38, 1e30, -1e30, 1e-45, 3.0]
    with built.Context(*dims) as ctx:
        for s in (1.2262736558914185, 3.0900158882141113):
            assert (bits(ctx.gauss_blur(vol, s)) == bits(oracle.blur(vol, s))).all()


@pytest.mark.parametrize("dims", [(32, 32, 32), (33, 21, 19), (64, 48, 40), (7, 5, 3)])
def test_dog_subsample_resize(built, oracle, dims):
    a, b = vol_of(built, dims, 1), vol_of(built, dims, 2)
    big = (2 * dims[0], 2 * dims[1], 2 * dims[2])
    with built.Context(*big) as ctx:
        assert (bits(ctx.dog(a, b)) == bits(oracle.dog(a, b))).all()
        if min(dims) >= 2:
            assert (bits(ctx.subsample2(a)) == bits(oracle.subsample(a))).all()
            assert (bits(ctx.halve_size(a)) == bits(oracle.halve(a))).all()
            assert (bits(ctx.double_size(a)) == bits(oracle.double_size(a))).all()


def _same_lists(got, want):
    assert len(got) == len(want)
    for f in ("x", "y", "z"):
        assert (got[f] == want[f]).all()
    assert (bits(got["value"]) == bits(want["value"])).all()


@pytest.mark.parametrize("dims", [(48, 40, 36), (65, 33, 20), (16, 16, 16), (3, 3, 3), (130, 9, 7)])
def test_extrema_lists(built, oracle, dims):
    vol = vol_of(built, dims, 21)
    g0 = oracle.blur(vol, 1.5198684930801392)
    G, D = oracle.octave_levels(g0)
    with built.Context(*dims) as ctx:
        for lvl in (1, 2, 3):
            mins, maxs = ctx.extrema(D[lvl - 1], D[lvl], D[lvl + 1])
            omin, omax = oracle.detect3(D[lvl - 1], D[lvl], D[lvl + 1])
            _same_lists(mins, omin)
            _same_lists(maxs, omax)
            mins, maxs = ctx.extrema(D[lvl - 1], D[lvl], None)   # the reference's GPU entry point: 26 + 27
            omin, omax = oracle.detect(D[lvl - 1], D[lvl])
            _same_lists(mins, omin)
            _same_lists(maxs, omax)


def test_extrema_dense_noise_grows_capacity(built, oracle):
    """White noise has ~1 extremum per 100 voxels per level: far above the blob density."""
    rng = np.random.default_rng(3)
    dims = (40, 40, 40)
    D = [rng.standard_normal(dims[::-1]).astype(np.float32) for _ in range(3)]
    with built.Context(*dims) as ctx:
        mins, maxs = ctx.extrema(D[0], D[1], D[2], capacity=dims[0] ** 3)
        omin, omax = oracle.detect3(D[0], D[1], D[2])
        _same_lists(mins, omin)
        _same_lists(maxs, omax)
        # plateaus: equal values are never strict extrema
        flat = np.zeros(dims[::-1], np.float32)
        mins, maxs = ctx.extrema(flat, flat, flat)
        assert len(mins) == 0 and len(maxs) == 0


@pytest.mark.parametrize("dims,seed,scale", [((64, 64, 64), 12345, 1.0), ((80, 64, 48), 777, 1.0), ((50, 47, 45), 8, 0.5)])
def test_pipeline_candidates(built, oracle, dims, seed, scale):
    vol = vol_of(built, dims, seed)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        got = ctx.detect(initial_image_scale=scale)
    want = oracle.candidates(vol, init_scale=scale)
    assert len(got) == len(want) > 0
    for f in ("octave", "level", "is_max", "x", "y", "z"):
        assert (got[f] == want[f]).all(), f
    for f in ("value", "h_value", "l_value"):
        assert (bits(got[f]) == bits(want[f])).all(), f


def _compare_records(got, want):
    assert len(got) == len(want)
    assert (got["info"] == want["info"]).all()
    for f in ("x", "y", "z", "scale", "ori", "eigs"):
        d = np.abs(got[f].astype(np.float64) - want[f].astype(np.float64))
        assert d.max() <= TOL, (f, d.max())
    assert (got["desc"] == want["desc"]).all()
    exact = all((bits(got[f]) == bits(want[f])).all() for f in ("x", "y", "z", "scale", "ori", "eigs"))
    return exact


@pytest.mark.parametrize("dims,seed", [((64, 64, 64), 12345), ((80, 64, 48), 777), ((96, 96, 96), 31)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_pipeline_records(built, oracle, dims, seed, mode):
    vol = vol_of(built, dims, seed)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        got = ctx.extract(desc_mode=mode)
    want, _ = oracle.extract(vol, desc_mode=mode)
    assert len(want) > 20
    exact = _compare_records(got, want)
    assert exact, "float fields are within 1e-4 but not bit-identical"


@pytest.mark.parametrize("dims,seed,mode", [((67, 45, 38), 5, 0), ((67, 45, 38), 5, 1), ((73, 90, 51), 21, 3)])
def test_pipeline_records_odd_dims(built, oracle, dims, seed, mode):
    """Rows that are not whole 16-byte vectors: padded to a pitch inside the pipeline, pad columns kept at zero."""
    vol = vol_of(built, dims, seed)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        got = ctx.extract(desc_mode=mode)
    want, _ = oracle.extract(vol, desc_mode=mode)
    assert len(want) > 5 and _compare_records(got, want)


def test_context_reuse_across_row_lengths(built, oracle):
    """Inside the pipeline rows are padded to whole 16-byte vectors and the pad columns must read as zero: reuse one
    context for an odd row length, a dense one, an operator-level call (which uses the level buffers as dense
    scratch) and the odd one again."""
    a_dims, b_dims = (67, 45, 38), (64, 48, 40)
    va, vb = vol_of(built, a_dims, 5), vol_of(built, b_dims, 12345)
    wa, _ = oracle.extract(va)
    wb, _ = oracle.extract(vb)
    with built.Context(68, 48, 40) as ctx:
        for vol, want in ((va, wa), (vb, wb), (va, wa)):
            ctx.set_volume(vol)
            assert _compare_records(ctx.extract(), want)
        junk = ctx.gauss_blur(vb + np.float32(100.0), 1.5450079441070557)   # dense scratch use of the level buffers
        assert (bits(junk) == bits(oracle.blur(vb + np.float32(100.0), 1.5450079441070557))).all()
        ctx.set_volume(va)
        assert _compare_records(ctx.extract(), wa)
        got = ctx.detect()
        want = oracle.candidates(va)
        assert len(got) == len(want) and (got["x"] == want["x"]).all() and (got["z"] == want["z"]).all()


@pytest.mark.parametrize("dims", [(100, 100, 100), (72, 72, 72), (168, 40, 36)])
def test_context_reuse_pitched_coarse_octaves(built, oracle, dims):
    """Row lengths that ARE whole 16-byte vectors but whose coarser octaves are not (100 -> 50 -> pitch 52, 72 -> 36 ->
    18 -> pitch 20, 168 -> 84 -> 42 -> pitch 44): the subsample writes the logical columns only, so the pad columns of a
    coarse octave's first level must be zeroed by the pipeline itself.  A larger, dense, strongly offset volume goes
    through the same context first so that those pad columns hold stale non-zero floats if the pipeline does not."""
    big = vol_of(built, (128, 128, 128), 3) + np.float32(500.0)
    vol = vol_of(built, dims, 21)
    want, _ = oracle.extract(vol)
    wc = oracle.candidates(vol)
    with built.Context(128, 128, 128) as ctx:
        ctx.set_volume(big)
        assert len(ctx.extract()) > 0
        ctx.set_volume(vol)
        got = ctx.extract()
        assert len(want) > 5 and _compare_records(got, want)
        gc = ctx.detect()
        assert len(gc) == len(wc) and all((gc[f] == wc[f]).all() for f in ("octave", "level", "is_max", "x", "y", "z"))
        assert (bits(gc["value"]) == bits(wc["value"])).all()


@pytest.mark.parametrize("dims", [(168, 164, 160), (166, 165, 161)])
def test_pipeline_records_through_the_fused_blur(built, oracle, dims):
    """A volume of more than 2^22 voxels: its finest octave is built by the one-launch ring kernel (two rows per thread;
    the 17-tap level is never filtered in full: test_lazy_levels_*), the 7- and 9-tap levels of the next octave (2^18
    voxels or more) too, with one row per thread; everything else by the three-pass kernels and the single-workgroup
    octave kernel.  The second shape has rows
    that are not whole 16-byte vectors (pitched rows).  Records against the oracle."""
    vol = vol_of(built, dims, 9)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        ctx.enable_timing(True)
        got = ctx.extract()
        log = ctx.launch_log()
    fused = log[log["stage"] == built.STAGES.index("blur_fused")]
    assert len(fused) == 7                    # octave 0: initial blur + four levels; octave 1: its 7- and 9-tap levels
    assert fused["ntaps"][:5].tolist() == [9, 7, 9, 11, 13]
    assert sorted(fused["ntaps"][5:].tolist()) == [7, 9] and (fused["nvox"][5:] < fused["nvox"][0]).all()
    want, _ = oracle.extract(vol)
    assert len(want) > 200 and _compare_records(got, want)


@pytest.mark.parametrize("dims,rows", [((168, 164, 160), 0), ((166, 165, 161), 0), ((88, 61, 47), 2), ((104, 96, 81), 2)])
def test_pipeline_level3_launch_carries_the_subsample(built, oracle, dims, rows):
    """The pyramid with the next octave's level 0 written by the level-3 blur launch (default where the octave is built by the
    two-rows-per-thread fused kernel and its pitched rows halve into whole vectors) against the separate subsample launch
    (TUNE_FUSED_SUB 0): the same records, which are the oracle's, and one subsample launch less per octave that carries.
    The small shapes force the fused kernel onto every octave it supports, so that coarse octaves with odd sizes carry too;
    (166, ...) has pitched rows (168) whose pad columns the launch must not let into the half-size volume."""
    vol = vol_of(built, dims, 13)
    with built.Context(*dims) as ctx:
        if rows:
            ctx.set_tuning(built.TUNE_BLUR_FUSED, 2)
            ctx.set_tuning(built.TUNE_FUSED_ROWS, rows)
        ctx.set_volume(vol)
        got = ctx.extract()
        st = ctx.timings()["stages"]
        ctx.set_tuning(built.TUNE_FUSED_SUB, 0)
        got0 = ctx.extract()
        st0 = ctx.timings()["stages"]
    assert got.tobytes() == got0.tobytes()
    assert st["subsample"]["launches"] < st0["subsample"]["launches"]
    assert st["blur_fused"]["launches"] == st0["blur_fused"]["launches"]
    want, _ = oracle.extract(vol)
    assert len(want) > 20 and _compare_records(got, want)


@pytest.mark.parametrize("dims,noise,mode", [((168, 164, 160), 0.0, 0), ((96, 80, 72), 0.0, 2), ((64, 64, 64), 30.0, 0), ((40, 36, 33), 0.0, 1),
                                             ((16, 16, 16), 0.0, 0), ((72, 72, 72), 6.0, 3)])
def test_split_tail_gives_the_same_records(built, oracle, dims, noise, mode):
    """Round 4: the candidate list in two parts (octaves 0 - 1 / the coarser ones), the first sorted and its keypoint kernel
    started while the coarse octaves are still being built (TUNE_SPLIT_TAIL 1, the default), against one sort and one launch
    behind the whole pyramid (0): the same bytes, the oracle's records.  Value 2 leaves the second part room for eight
    extrema: it overflows while the first part's keypoint kernel is in flight and the run falls back to the one-list
    schedule with a replay; the dense-noise volume overflows the first part (and the own-level lists) instead.  The context is
    used again afterwards with the default, so a fall-back leaves nothing behind."""
    vol = vol_of(built, dims, 41)
    if noise:
        vol = vol + (np.random.default_rng(3).standard_normal(vol.shape) * noise).astype(np.float32)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        runs = {}
        for v in (1, 0, 2, 1):
            ctx.set_tuning(built.TUNE_SPLIT_TAIL, v)
            got = ctx.extract(desc_mode=mode)
            tm = ctx.timings()
            runs.setdefault(v, []).append((got.tobytes(), tm["n_extrema"], tm["n_records"], tm["stages"]["keypoint"]["launches"]))
            last = got.copy()
    ref = runs[0][0]
    for v, rs in runs.items():
        for r in rs:
            assert r[:3] == ref[:3], v
    want, _ = oracle.extract(vol, desc_mode=mode)
    assert len(want) == ref[2] and (len(want) == 0 or _compare_records(last, want))
    assert runs[0][0][3] <= 1
    if dims[0] >= 160:
        assert runs[1][0][3] == 2    # two keypoint launches: the split schedule ran (a part without extrema has none)


@pytest.mark.parametrize("dims,mode", [((96, 80, 72), 0), ((120, 100, 90), 3), ((40, 36, 33), 1)])
def test_descriptor_record_order_over_the_xcds_changes_nothing(built, dims, mode):
    """Round 4: which workgroup of the descriptor kernel takes which record (TUNE_DESC_SEGMENT: contiguous eighths of segments of
    8 n records per XCD; 0 = of the whole list, 1 = round-robin) only moves work between the XCDs: the same bytes for every
    segment length, also one that is longer than the list and one that leaves a partial last segment."""
    vol = vol_of(built, dims, 17)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract(desc_mode=mode)
        assert len(want) > 3
        for seg in (0, 1, 3, 7, 32, 1000, 1 << 20):
            ctx.set_tuning(built.TUNE_DESC_SEGMENT, seg)
            assert ctx.extract(desc_mode=mode).tobytes() == want.tobytes(), seg


def test_pipeline_empty_volume(built, oracle, tmp_path):
    """No extremum anywhere: zero records, and the CLI still writes a well-formed .key."""
    dims = (48, 40, 36)
    vol = np.zeros(dims[::-1], np.float32)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        assert len(ctx.detect()) == 0
        assert len(ctx.extract()) == 0
    want, _ = oracle.extract(vol)
    assert len(want) == 0
    nii, k = str(tmp_path / "zero.nii"), str(tmp_path / "zero.key")
    built.write_nifti(nii, vol)
    r = subprocess.run([built.FEATEXTRACT, "-d0", nii, k], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(k).read().splitlines()
    assert lines[0] == "# featExtract 1.1" and "Features: 0" in lines and len(lines) == 6


def test_pipeline_double_and_halve(built, oracle):
    dims = (40, 36, 32)
    vol = vol_of(built, dims, 99)
    with built.Context(2 * dims[0], 2 * dims[1], 2 * dims[2]) as ctx:
        big = ctx.double_size(vol)
        assert (bits(big) == bits(oracle.double_size(vol))).all()
        ctx.set_volume(big)
        got = ctx.extract(initial_image_scale=0.5, size_factor=0.5)
        want, _ = oracle.extract(big, init_scale=0.5, size_factor=0.5)
        assert len(want) > 5 and _compare_records(got, want)
        ctx.set_volume(vol, resize=+1)
        again = ctx.extract(initial_image_scale=0.5, size_factor=0.5)
        assert (again.view(np.uint8) == got.view(np.uint8)).all()
    big_dims = (96, 80, 72)
    vol = vol_of(built, big_dims, 5)
    with built.Context(*big_dims) as ctx:
        small = ctx.halve_size(vol)
        ctx.set_volume(small)
        got = ctx.extract(size_factor=2.0)
        want, _ = oracle.extract(oracle.halve(vol), size_factor=2.0)
        assert len(want) > 5 and _compare_records(got, want)
        # the same resize between the upload and the pyramid, without the trip through the host (what the CLI uses)
        ctx.set_volume(vol, resize=-1)
        again = ctx.extract(size_factor=2.0)
        assert (again.view(np.uint8) == got.view(np.uint8)).all()


def test_cli_key_file_is_byte_identical(built, oracle, tmp_path):
    """featExtract -d0 on the GPU == the CPU restatement's .key, byte for byte, and the golden fixture."""
    import _oracle
    vol = vol_of(built, (64, 64, 64), 12345)
    nii = str(tmp_path / "in.nii")
    built.write_nifti(nii, vol)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for flag, name in ((None, "sift"), ("-b", "brief"), ("-br", "rrief"), ("-bn", "nrrief")):
        k = str(tmp_path / ("gpu_%s.key" % name))
        cmd = [built.FEATEXTRACT, "-d0"] + ([flag] if flag else []) + [nii, k]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Extracting features: %s" % nii in r.stdout and "Input image: i=64 j=64 k=64" in r.stdout
        assert r.stdout.endswith("\nDone.\n")
        assert open(k, "rb").read() == open(os.path.join(gold, "oracle_blob64_%s.key" % name), "rb").read()
    # -2+ through the CLI against the oracle CLI
    k1, k2 = str(tmp_path / "a.key"), str(tmp_path / "b.key")
    small = str(tmp_path / "s.nii")
    built.write_nifti(small, vol_of(built, (40, 36, 32), 99))
    assert subprocess.run([built.FEATEXTRACT, "-2+", "-d0", small, k1], capture_output=True).returncode == 0
    assert subprocess.run([_oracle.CLI, "-2+", small, k2], capture_output=True).returncode == 0
    assert open(k1, "rb").read() == open(k2, "rb").read()
    # -2- likewise
    assert subprocess.run([built.FEATEXTRACT, "-2-", "-d0", nii, k1], capture_output=True).returncode == 0
    assert subprocess.run([_oracle.CLI, "-2-", nii, k2], capture_output=True).returncode == 0
    assert open(k1, "rb").read() == open(k2, "rb").read() and len(open(k1).readlines()) > 8


def test_volume_in_runs_of_planes_is_the_volume(built):
    """sift3d_set_volume_begin / _planes / _end (round 5: the CLI uploads what it has read while the rest of the file is still
    being inflated): dense and pitched rows, -2+ and -2-, runs in and out of order -- the records of sift3d_set_volume; a plane
    missing, a plane beyond the volume, planes without a begin: refused."""
    for dims, resize in (((64, 48, 40), 0), ((50, 44, 37), 0), ((40, 36, 32), 1), ((64, 64, 64), -1), ((33, 30, 41), 1)):
        vol = vol_of(built, dims, 17)
        nz = dims[2]
        cd = tuple(2 * d for d in dims) if resize > 0 else dims
        with built.Context(*cd) as ctx:
            ctx.set_volume(vol, resize=resize)
            want = ctx.extract()
            for runs in ([(0, nz)], [(0, 7), (7, 1), (8, nz - 8)], [(nz - 5, 5), (0, 11), (11, nz - 16)]):
                ctx.set_volume_in_runs(vol, runs, resize=resize)
                got = ctx.extract()
                assert got.tobytes() == want.tobytes() and len(got) >= 5, (dims, resize, runs)
            with pytest.raises(built.Sift3DError):
                ctx.set_volume_in_runs(vol, [(0, nz - 1)], resize=resize)          # a plane never arrived
            with pytest.raises(built.Sift3DError):
                ctx.set_volume_in_runs(vol, [(0, nz), (nz - 1, 2)], resize=resize)   # beyond the volume
            with pytest.raises(built.Sift3DError):                                    # nz planes in all, but three of them twice and
                ctx.set_volume_in_runs(vol, [(0, nz - 6), (nz - 9, 6)], resize=resize)   # the last three never (round-5 advisor finding)
            with pytest.raises(built.Sift3DError):
                ctx._chk(ctx._L.sift3d_set_volume_planes(ctx._h, vol.ctypes.data, 0, 1), "planes without begin")
            ctx.set_volume(vol, resize=resize)                                        # the context is still usable
            assert ctx.extract().tobytes() == want.tobytes()


def test_cli_reads_and_uploads_side_by_side(built, tmp_path):
    """The command line brings up the device and uploads the planes it has while the file is still being read (round 5): the
    same .key for .nii, .nii.gz and an int16 file of several runs of planes, with and without -2+; a truncated file is the
    reference's 'could not read input file', not a hang of the uploading thread."""
    import gzip
    import _oracle
    vol = np.round(vol_of(built, (96, 80, 300), 5))          # 2.3 M voxels a ... 9 MB: more than one 32 MB run only as int16? no: force several runs below
    vol = np.tile(vol, (4, 1, 1))[:1100]                     # 96 x 80 x 1100 = 8.4 M voxels = 34 MB: two runs
    nii, gz, i16 = str(tmp_path / "v.nii"), str(tmp_path / "v.nii.gz"), str(tmp_path / "i16.nii")
    built.write_nifti(nii, vol)
    with open(nii, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
        g.write(f.read())
    raw = bytearray(open(nii, "rb").read()[:352])
    raw[70:72] = (4).to_bytes(2, "little"); raw[72:74] = (16).to_bytes(2, "little")   # datatype DT_INT16, bitpix 16
    with open(i16, "wb") as f:
        f.write(bytes(raw) + vol.astype("<i2").tobytes())
    keys = {}
    for src in (nii, gz, i16):
        for flags in ([], ["-2-"]):
            k = str(tmp_path / "out.key")
            r = subprocess.run([built.FEATEXTRACT, "-d0"] + flags + [src, k], capture_output=True, text=True, env=dict(os.environ, SIFT3D_CLI_TIMES="1"))
            assert r.returncode == 0, r.stdout + r.stderr
            assert "the planes were uploaded while the file was read" in r.stderr
            keys.setdefault(tuple(flags), []).append(open(k, "rb").read())
    for flags, ks in keys.items():
        assert ks[0] == ks[1] == ks[2] and ks[0].count(b"\n") > 50, flags
    k2 = str(tmp_path / "cpu.key")
    assert subprocess.run([_oracle.CLI, nii, k2], capture_output=True).returncode == 0
    assert open(k2, "rb").read() == keys[()][0]
    cut = str(tmp_path / "cut.nii")
    with open(cut, "wb") as f:
        f.write(open(nii, "rb").read()[:352 + 4 * 96 * 80 * 700])
    r = subprocess.run([built.FEATEXTRACT, "-d0", cut, str(tmp_path / "x.key")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 255 and "Error: could not read input file" in r.stdout


def test_cli_side_effects_of_the_reference(built, oracle, tmp_path):
    """What the reference's pyramid leaves behind besides its result (MultiScale.cpp:296-302,373-388,558): a '#<microseconds>'
    line after the initial blur and after the first blur of every octave, 'done.' per octave, and ./image.pgm -- the middle
    slice of octave 0's first blurred level, scaled to 0..255 (checked here against the oracle's level)."""
    dims = (64, 56, 48)
    vol = vol_of(built, dims, 17)
    nii = str(tmp_path / "in.nii")
    built.write_nifti(nii, vol)
    r = subprocess.run([built.FEATEXTRACT, "-d0", nii, str(tmp_path / "o.key")], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.split("\n")
    n_oct = 5                                                            # 64 x 56 x 48 halves to 4 x 3 x 3
    timers = [l for l in lines if l.startswith("#")]
    assert len(timers) == 1 + n_oct and all(l[1:].isdigit() for l in timers)
    assert lines.count("done.") == n_oct
    assert lines.index("Input image: i=64 j=56 k=48") < lines.index(timers[0]) and r.stdout.endswith("\nDone.\n")
    pgm = open(tmp_path / "image.pgm", "rb").read()
    head = b"P5\n64 56\n255\n"
    assert pgm.startswith(head) and len(pgm) == len(head) + 64 * 56
    l0 = oracle.blur(vol, 1.5198684930801392)
    l1 = oracle.blur(l0, 1.2262736558914185)[dims[2] // 2]
    lo, hi = np.float32(l1.min()), np.float32(l1.max())
    want = (((l1 - lo).astype(np.float64) * 255.0) / np.float64(hi - lo)).astype(np.uint8)
    assert (np.frombuffer(pgm[len(head):], np.uint8).reshape(56, 64) == want).all()


@pytest.mark.parametrize("flag", ["-w", "-ws"])
def test_cli_world_coordinates(built, tmp_path, flag):
    """-w / -ws on an anisotropic volume with distinct qform and sform: byte-identical to the oracle's .key."""
    import _oracle
    w = _oracle.WORLD_CASE
    nii, k = str(tmp_path / "aniso.nii"), str(tmp_path / "gpu.key")
    built.write_nifti(nii, vol_of(built, w["dims"], w["seed"]), w["voxel"], w["qform"], w["sform"])
    r = subprocess.run([built.FEATEXTRACT, flag, "-d0", nii, k], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Input image: i=96 j=100 k=84" in r.stdout
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_aniso_%s.key" % flag[1:])
    assert open(k, "rb").read() == open(gold, "rb").read()


def test_cli_reproduces_the_shipped_binary(built, tmp_path):
    """The HIP path against the reference's OWN output, byte for byte: tests/golden/refbin_*.key are the .key files the CPU binary
    under the reference's bin/Linux wrote (round 1; frozen data) for the 64^3 and 128^3 blob fields and for the anisotropic -w / -ws
    case.  That binary's toolchain evaluated exp() of a float with the C exp(double) (its disassembly; include/sift3d.h,
    sift3d_set_libm_variant); `featExtract --libm=gcc5` builds the Gaussian taps that way and must then write the binary's files
    exactly -- 74, 1 698 and 2 x 180-odd records of 81 printed columns, positions, scales, frames, eigenvalues, flags and rank
    descriptors.  Without the switch (the reference as a current g++ compiles it) the same runs stay within the tolerances of
    tests/test_oracle_pins.py::test_against_shipped_reference_binary, which is the last bit of two taps carried through."""
    import gzip
    import _oracle
    from keyio import read_key
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for n, name in ((64, "refbin_blob64.key"), (128, "refbin_blob128.key.gz")):
        nii, k, k0 = str(tmp_path / ("b%d.nii" % n)), str(tmp_path / ("b%d.key" % n)), str(tmp_path / ("b%d_default.key" % n))
        built.write_nifti(nii, vol_of(built, (n, n, n), 12345))
        r = subprocess.run([built.FEATEXTRACT, "--libm=gcc5", "-d0", nii, k], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        path = os.path.join(gold, name)
        want = gzip.open(path, "rb").read() if name.endswith(".gz") else open(path, "rb").read()
        assert open(k, "rb").read() == want, name
        # the default build of the taps: same records, the reference binary's numbers within the north star's 1e-4
        assert subprocess.run([built.FEATEXTRACT, "-d0", nii, k0], capture_output=True).returncode == 0
        a, b = read_key(k0), read_key(gzip.open(path, "rt") if name.endswith(".gz") else path)
        ra, rb = a["rows"], b["rows"]
        assert a["count"] == b["count"] == len(ra) == len(rb) and (ra[:, 16] == rb[:, 16]).all()
        d = np.abs(ra - rb)
        assert d[:, :4].max() < 2e-4 and (d[:, 13:16] / np.abs(rb[:, 13:16])).max() < 2e-4
        assert (d[:, 17:].max(1) == 0).mean() >= 0.995 and d[:, 17:].max() <= 2
        assert np.minimum(np.abs(ra[:, 4:13] - rb[:, 4:13]), np.abs(ra[:, 4:13] + rb[:, 4:13])).max() < 2e-3
    w = _oracle.WORLD_CASE
    nii = str(tmp_path / "aniso.nii")
    built.write_nifti(nii, vol_of(built, w["dims"], w["seed"]), w["voxel"], w["qform"], w["sform"])
    for flag in ("-w", "-ws"):
        k = str(tmp_path / ("aniso%s.key" % flag))
        r = subprocess.run([built.FEATEXTRACT, "--libm=gcc5", flag, "-d0", nii, k], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(k, "rb").read() == open(os.path.join(gold, "refbin_aniso_%s.key" % flag[1:]), "rb").read(), flag
    # -2+ with the binary's taps: the 224 records the round-1 review counted from the shipped binary on this volume (223 with the
    # default taps: one near-tie of an orientation peak resolves the other way), byte-identical to the oracle's variant build
    b64, kg, kv = str(tmp_path / "b64.nii"), str(tmp_path / "g2.key"), str(tmp_path / "v2.key")
    assert subprocess.run([built.FEATEXTRACT, "--libm=gcc5", "-2+", "-d0", b64, kg], capture_output=True).returncode == 0
    assert subprocess.run([_oracle.CLI_REFBIN, "-2+", b64, kv], capture_output=True).returncode == 0
    assert open(kg, "rb").read() == open(kv, "rb").read() and read_key(kg)["count"] == 224
    # an unknown long option is an unknown argument, as every "--..." is for the reference
    r = subprocess.run([built.FEATEXTRACT, "--libm=gcc4", "-d0", nii, str(tmp_path / "x.key")], capture_output=True, text=True)
    assert r.returncode == 255 and "Error: unknown command line argument: --libm=gcc4" in r.stdout


def test_cli_ws_without_sform_falls_back_to_qform(built, tmp_path):
    import _oracle
    w = _oracle.WORLD_CASE
    nii, k1, k2 = str(tmp_path / "q.nii"), str(tmp_path / "a.key"), str(tmp_path / "b.key")
    built.write_nifti(nii, vol_of(built, (64, 48, 40), 7), w["voxel"], w["qform"], None)
    r = subprocess.run([built.FEATEXTRACT, "-ws", "-d0", nii, k1], capture_output=True, text=True)
    assert r.returncode == 0 and "Error: sform_code <= 0, output to qto_xyz instead of sto_xyz" in r.stdout
    assert subprocess.run([_oracle.CLI, "-ws", nii, k2], capture_output=True).returncode == 0
    assert open(k1, "rb").read() == open(k2, "rb").read()


def test_full_size_properties(built):
    """BASELINE config sizes the oracle cannot finish quickly: size-independent checks at 256^3."""
    n = 256
    vol = vol_of(built, (n, n, n), 12345)
    with built.Context(n, n, n) as ctx:
        # linearity of the blur in its input under exact scalings: blur(2v) == 2 blur(v)
        a = ctx.gauss_blur(vol, 3.0900158882141113)
        b = ctx.gauss_blur(vol * np.float32(2), 3.0900158882141113)
        assert (bits(b) == bits(a * np.float32(2))).all()
        # mirror symmetry: taps are symmetric only up to summation order, so compare with tolerance
        c = ctx.gauss_blur(vol[::-1, ::-1, ::-1].copy(), 3.0900158882141113)[::-1, ::-1, ::-1]
        assert np.abs(c - a).max() <= 1e-3 * np.abs(a).max()
        # a constant volume stays constant away from the zero-padded border
        k = ctx.gauss_blur(np.full((n, n, n), 7.0, np.float32), 1.9465880393981934)
        core = k[8:-8, 8:-8, 8:-8]
        assert core.min() == core.max() and abs(float(core[0, 0, 0]) - 7.0) < 1e-5
        ctx.set_volume(vol)
        cands = ctx.detect()
        # candidates come in the reference's order: octave, level, minima before maxima, raster index
        key = (cands["octave"].astype(np.int64) << 40) + (cands["level"].astype(np.int64) << 36) + (cands["is_max"].astype(np.int64) << 32)
        lin = (cands["z"].astype(np.int64) * 512 + cands["y"]) * 512 + cands["x"]
        order = np.lexsort((lin, key))
        assert (order == np.arange(len(cands))).all()
        assert ((cands["is_max"] == 1) == (cands["value"] > cands["h_value"])).all()
        assert len(cands) == 5546  # validated extrema of the oracle at 256^3 (SURVEY generator, seed 12345)
        feats = ctx.extract()
        assert len(feats) == 19216  # record count of the source-built reference (tests/golden/ref_counts.json)
        # idempotence: a second run on the same context gives the same records
        again = ctx.extract()
        assert (again.view(np.uint8) == feats.view(np.uint8)).all()


def test_config_c2_against_the_oracle(built, oracle):
    """BASELINE config C2 at its own size -- 256^3 blob field (the SURVEY generator, seed 12345), every octave, SIFT-rank
    descriptor -- record by record against the CPU restatement (about 7 s of one host core): integer fields and rank
    descriptors exact, float fields within 1e-4 (north_star) and, as everywhere in this suite, bit-identical."""
    n = 256
    vol = vol_of(built, (n, n, n), 12345)
    with built.Context(n, n, n) as ctx:
        ctx.set_volume(vol)
        got = ctx.extract()
        cands = ctx.detect()
    want, st = oracle.extract(vol)
    assert len(want) == 19216 and st.n_octaves == 7          # the source-built reference's count (tests/golden/ref_counts.json)
    assert _compare_records(got, want), "float fields are within 1e-4 but not bit-identical"
    wc = oracle.candidates(vol)
    assert len(cands) == len(wc) == 5546
    for f in ("octave", "level", "is_max", "x", "y", "z"):
        assert (cands[f] == wc[f]).all(), f
    for f in ("value", "h_value", "l_value"):
        assert (bits(cands[f]) == bits(wc[f])).all(), f


def test_pinned_record_buffers_grow_when_a_run_needs_more(built):
    """advisor, round 3: the pinned download buffers are sized for a few records per candidate (5; blob fields yield 4.2), not
    for the worst case of 12.  A run that yields more grows them between the keypoint and the descriptor kernel -- forced here
    by sizing them for ONE record per candidate -- and returns the same bytes, in one chunk and in several (where records of
    chunks already described are carried over), on a fresh and on a reused context."""
    dims = (192, 176, 160)
    vol = vol_of(built, dims, 13)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract()
        assert len(want) > 4000 and ctx.host_buffer_grows() == 0
        assert len(want) > 1.125 * ctx.timings()["n_extrema"] + 1024     # 1 per candidate (plus the allocation's slack) cannot hold them
    for chunks in (0, 1, 3, 8):     # 0: the default schedule (two parts, the split tail)
        with built.Context(*dims) as ctx:
            ctx.set_tuning(built.TUNE_HOST_RECORDS, 1)
            ctx.set_tuning(built.TUNE_KP_CHUNKS, chunks)
            ctx.set_volume(vol)
            got = ctx.extract()
            assert ctx.host_buffer_grows() >= 1 and got.tobytes() == want.tobytes(), chunks
            grown = ctx.host_buffer_grows()
            assert ctx.extract().tobytes() == want.tobytes() and ctx.host_buffer_grows() == grown   # now large enough
            assert ctx.extract(copy=False).tobytes() == want.tobytes()



@pytest.mark.parametrize("dims,mode", [((168, 164, 160), 0), ((200, 120, 96), 2)])
def test_chunked_keypoint_stage_gives_the_same_records(built, dims, mode):
    """The per-keypoint stage cut into chunks (keypoint kernel of chunk i+1 beside the descriptor kernel of chunk i, on two
    streams) against one launch of each kernel: the same bytes for every chunk count, also when chunks are empty, on a reused
    context, with every launch bracketed by timing events, and with the descriptor kernel's sampling limit off."""
    vol = vol_of(built, dims, 13)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        ctx.set_tuning(built.TUNE_KP_CHUNKS, 1)
        want = ctx.extract(desc_mode=mode)
        t = ctx.timings()
        assert len(want) > 1000 and t["stages"]["keypoint"]["launches"] == 1 and t["stages"]["descriptor"]["launches"] == 1
        for chunks in (2, 3, 7, 16, 0):
            ctx.set_tuning(built.TUNE_KP_CHUNKS, chunks)
            for _ in range(2):
                assert ctx.extract(desc_mode=mode).tobytes() == want.tobytes(), chunks
            if chunks:
                assert ctx.timings()["stages"]["keypoint"]["launches"] == chunks
        ctx.set_tuning(built.TUNE_KP_CHUNKS, 4)
        ctx.enable_timing(1)
        assert ctx.extract(desc_mode=mode).tobytes() == want.tobytes()
        assert ctx.timings()["stages"]["descriptor"]["ms"] > 0
        ctx.enable_timing(0)
        ctx.set_tuning(built.TUNE_SAMPLER_CAP, 0)
        assert ctx.extract(desc_mode=mode).tobytes() == want.tobytes()
    with built.Context(*dims) as ctx:                                   # fresh context, chunked from the first run
        ctx.set_tuning(built.TUNE_KP_CHUNKS, 5)
        ctx.set_volume(vol)
        assert ctx.extract(desc_mode=mode).tobytes() == want.tobytes()
    tiny = vol_of(built, (24, 20, 18), 3)                               # fewer candidates than chunks
    with built.Context(24, 20, 18) as ctx:
        ctx.set_volume(tiny)
        ctx.set_tuning(built.TUNE_KP_CHUNKS, 1)
        w = ctx.extract()
        ctx.set_tuning(built.TUNE_KP_CHUNKS, 16)
        assert ctx.extract().tobytes() == w.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("dims,noise", [((168, 164, 160), 0.0), ((96, 50, 67), 4.0), ((40, 36, 33), 0.0), ((9, 70, 64), 1.0)])
def test_lazy_levels_give_the_same_candidates_and_records(built, oracle, dims, noise):
    """The default pipeline stores neither D_0 nor D_4 nor L_5 (the level below D_1 is taken as L_0 - L_1 around the
    extrema, the level above D_3 is filtered only in the 27-voxel neighbourhood of what passed every other test);
    sift3d_set_tuning(SIFT3D_TUNE_LAZY_LEVELS, 0) stores and filters everything as the reference does.  Candidates (with
    the DoG values one level
    below and above) and records are the same bytes both ways, and the candidates are the oracle's."""
    vol = vol_of(built, dims, 29)
    if noise:
        vol = vol + (np.random.default_rng(5).standard_normal(vol.shape) * noise).astype(np.float32)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        cand = ctx.detect()
        recs = ctx.extract()
        stages = ctx.timings()["stages"]
        ctx.set_tuning(built.TUNE_LAZY_LEVELS, 0)
        cand_full = ctx.detect()
        recs_full = ctx.extract()
        stages_full = ctx.timings()["stages"]
        ctx.set_tuning(built.TUNE_LAZY_LEVELS, 1)
        assert ctx.extract().tobytes() == recs.tobytes()
    assert len(cand) > 5
    assert cand.tobytes() == cand_full.tobytes()
    assert recs.tobytes() == recs_full.tobytes()
    want = oracle.candidates(vol)
    assert len(want) == len(cand)
    for f in ("octave", "level", "is_max", "x", "y", "z"):
        assert (cand[f] == want[f]).all(), f
    for f in ("value", "h_value", "l_value"):
        assert (cand[f].view(np.uint32) == want[f].view(np.uint32)).all(), f
    # fewer blur launches on every octave that is not a single-workgroup one
    n_blur = lambda st: sum(st[k]["launches"] for k in ("blur_fused", "blur_x"))
    assert n_blur(stages) < n_blur(stages_full)


def test_octave_limit_returns_the_leading_records(built, oracle):
    """sift3d_set_max_octaves: records and candidates are octave-major, so a run limited to n octaves is a prefix of the
    unlimited run (which is the oracle's)."""
    dims = (128, 112, 96)
    vol = vol_of(built, dims, 6)
    want, _ = oracle.extract(vol)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        full, cfull = ctx.extract(), ctx.detect()
        top = int(cfull["octave"].max())                      # the coarsest octave that still has validated extrema
        assert _compare_records(full, want) and top >= 1
        for n in range(1, top + 3):
            ctx.set_max_octaves(n)
            c = ctx.detect()
            keep = cfull[cfull["octave"] < n]
            assert len(c) == len(keep) and c.tobytes() == keep.tobytes()
            f = ctx.extract()
            assert ctx.timings()["n_octaves"] == min(n, 6)    # 128 x 112 x 96 has six octaves by the reference's stop rule
            assert 0 < len(f) <= len(full) and (len(f) < len(full)) == (n <= top)
            assert f.tobytes() == full[:len(f)].tobytes()
        ctx.set_max_octaves(0)
        assert ctx.extract().tobytes() == full.tobytes()


def _record_properties(f, dims, rank_desc=True):
    """Size-independent properties of a record list (used where the oracle cannot run in test time)."""
    nx, ny, nz = dims
    for k in ("x", "y", "z", "scale", "ori", "eigs"):
        assert np.isfinite(f[k]).all(), k
    assert (f["x"] >= 0).all() and (f["x"] <= nx).all() and (f["y"] >= 0).all() and (f["y"] <= ny).all()
    assert (f["z"] >= 0).all() and (f["z"] <= nz).all() and (f["scale"] > 0).all()
    assert ((f["info"] & ~np.uint32(0x30)) == 0).all()                      # only the min/max and reorient flags
    if rank_desc:                                                           # NormalizeDataRankedPCs: a permutation of 0..63
        assert (np.sort(f["desc"], axis=1) == np.arange(64, dtype=np.float32)).all()
    # every keypoint is one un-reoriented record followed by its reoriented frames at the same position
    first = (f["info"] & np.uint32(0x20)) == 0
    assert first[0] and first.sum() > 0
    same_as_prev = (f["x"][1:] == f["x"][:-1]) & (f["y"][1:] == f["y"][:-1]) & (f["z"][1:] == f["z"][:-1]) & (f["scale"][1:] == f["scale"][:-1])
    assert same_as_prev[~first[1:]].all()
    # the eigenvalue test every record passed (featExtract.cpp:297, MultiScale.cpp:1748-1769), in float as there
    e = f["eigs"].astype(np.float32)
    ssum = (e[:, 0] + e[:, 1] + e[:, 2]).astype(np.float32)
    assert ((ssum * ssum * ssum) < np.float32(140.0) * (e[:, 0] * e[:, 1] * e[:, 2])).mean() > 0.999
    # rows of an orientation frame are unit vectors
    o = f["ori"].reshape(-1, 3, 3).astype(np.float64)
    assert np.abs(np.linalg.norm(o, axis=2) - 1.0).max() < 1e-3


def test_config_c3_flags_bit_exact(built, oracle):
    """BASELINE config C3's options -- -2+ (fioDoubleSize on the device, initial image scale 0.5, size factor 0.5:
    featExtract.cpp:368-376,423-427) and the BRIEF descriptor -- at a size the oracle finishes: 96^3 -> 192^3."""
    dims = (96, 96, 96)
    vol = vol_of(built, dims, 31)
    want, _ = oracle.extract(oracle.double_size(vol), init_scale=0.5, desc_mode=1, size_factor=0.5)
    with built.Context(192, 192, 192) as ctx:
        ctx.set_volume(vol, resize=+1)
        got = ctx.extract(initial_image_scale=0.5, desc_mode=built.DESC_BRIEF, size_factor=0.5)
        assert len(want) > 200 and _compare_records(got, want)
        ctx.set_volume(vol, resize=-1)                                       # -2-: fioSubSample2DCenterPixel, size factor 2
        got = ctx.extract(initial_image_scale=1.0, desc_mode=built.DESC_BRIEF, size_factor=2.0)
    want, _ = oracle.extract(oracle.halve(vol), init_scale=1.0, desc_mode=1, size_factor=2.0)
    assert _compare_records(got, want)


def _oracle_on_all_cores(done="the properties above"):
    """The OpenMP build of the CPU restatement, for the full-size configurations: its records are the serial library's byte
    for byte (tests/test_oracle_pins.py holds it to that), and it finishes a 2^30-voxel volume in about a minute where the
    serial build needs three.  On a box with less than 80 GB of free host memory the test is reported as SKIPPED with the
    reason (round-3 review: it used to pass silently without the oracle comparison) -- the callers therefore make this
    their last step, after every property that needs no oracle."""
    import psutil
    avail = psutil.virtual_memory().available
    if avail < 80 * 2 ** 30:
        pytest.skip("oracle comparison NOT run: %.0f GB of free host memory, the full-size CPU restatement needs 80 "
                    "(%s passed)" % (avail / 2 ** 30, done))
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
    import _oracle
    return _oracle.load_omp()


def test_config_c3_full_size(built):
    """BASELINE config C3 at its own size: 512^3 float32, -2+ (processing size 1024^3 = 2^30 voxels), BRIEF, one GPU:
    size-independent properties, idempotence, and -- round 3 -- every record against the CPU restatement run on all host
    cores (about a minute and 40 GB of host memory)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2 ** 30:
        pytest.skip("needs about 70 GB of free HBM")
    n = 512
    vol = vol_of(built, (n, n, n), 12345)
    with built.Context(2 * n, 2 * n, 2 * n) as ctx:
        ctx.set_volume(vol, resize=+1)
        f = ctx.extract(initial_image_scale=0.5, desc_mode=built.DESC_BRIEF, size_factor=0.5)
        t = ctx.timings()
        assert t["n_octaves"] == 9 and t["n_extrema"] > 50000 and len(f) > 200000
        _record_properties(f, (n, n, n))
        f = f.copy()
        again = ctx.extract(initial_image_scale=0.5, desc_mode=built.DESC_BRIEF, size_factor=0.5)
        assert (again.view(np.uint8) == f.view(np.uint8)).all()             # idempotent
        # the SIFT-rank run of the same volume finds the same keypoints (the descriptor mode only changes desc)
        g = ctx.extract(initial_image_scale=0.5, desc_mode=built.DESC_SIFT, size_factor=0.5)
        assert len(g) == len(f)
        for k in ("x", "y", "z", "scale", "ori", "eigs", "info"):
            assert np.ascontiguousarray(g[k]).tobytes() == np.ascontiguousarray(f[k]).tobytes(), k
        assert (g["desc"] != f["desc"]).any()
    orc = _oracle_on_all_cores("size-independent properties, idempotence and the descriptor-mode check")   # last: may skip
    want, _ = orc.extract(orc.double_size(vol), init_scale=0.5, desc_mode=1, size_factor=0.5)
    assert _compare_records(f, want), "float fields are within 1e-4 but not bit-identical"


def test_config_c4_volume_single_gpu(built):
    """The volume of BASELINE config C4 (1024 x 1024 x 512, 2^29 voxels) on one GPU: properties, idempotence and -- round 3 --
    every record against the CPU restatement run on all host cores.  The four-slab split of the same volume is the next test;
    its merged records must be these bytes."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs about 40 GB of free HBM")
    dims = (1024, 1024, 512)
    vol = vol_of(built, dims, 12345)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        f = ctx.extract()
        t = ctx.timings()
        assert t["n_octaves"] == 8 and len(f) > 500000
        _record_properties(f, dims)
        f = f.copy()
        again = ctx.extract()
        assert (again.view(np.uint8) == f.view(np.uint8)).all()
        cands = ctx.detect()
        key = (cands["octave"].astype(np.int64) << 40) + (cands["level"].astype(np.int64) << 36) + (cands["is_max"].astype(np.int64) << 32)
        lin = (cands["z"].astype(np.int64) * 1024 + cands["y"]) * 1024 + cands["x"]
        assert (np.lexsort((lin, key)) == np.arange(len(cands))).all()      # the reference's order
        assert len(cands) == t["n_extrema"]
    orc = _oracle_on_all_cores("size-independent properties, idempotence and the order of the candidates")   # last: may skip
    want, _ = orc.extract(vol)
    assert _compare_records(f, want), "float fields are within 1e-4 but not bit-identical"


def _tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_randomized_parity_sweep(built, oracle):
    """40 random (shape, seed, noise, descriptor mode, initial scale, blur path) cases, records bit-identical to the
    oracle's (tools/fuzz_parity.py, which since round 3 also draws the chunk count of the per-keypoint stage; 200 cases of the same sweep, seed 2027, and 300 with seed 2028 ran clean on the round-3 build, 600 with seed 2026 on round 2's;
    round 4, with the fused blur's tile and the pinned buffers' starting size drawn too: 300, 1 500 and 4 000 cases, seeds 2029 - 2031, clean)."""
    assert _tool("fuzz_parity").sweep(40, 11, 112) == 0


def test_volume_beyond_32_bit_indices(built):
    """Maximum sizes: 1280 x 1280 x 1408 = 2.3e9 voxels (linear indices beyond 2^31, byte offsets beyond 2^33), which
    the reference's int arithmetic cannot address.  Translation property instead of an oracle run:
    a blob block near the far corner gives the candidates of the same block in a small volume, shifted, with
    bit-identical DoG values; see tools/big_volume_check.py."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * 2 ** 30:
        pytest.skip("needs about 160 GB of free HBM")
    assert _tool("big_volume_check").check()


# ---------------------------------------------------------------------------------------------------
# Z-slab extraction driven from C: one process, one context per listed device, halos by hipMemcpyPeerAsync.  On the
# one-GPU test box the same device is listed several times: the slab logic (plan, halo widths, thin + deferred exchange,
# own-slice filtering, coarse-octave gather, merge order) runs exactly as on several devices, the copies are device-local.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims,seed,mode,devs,scale", [((96, 80, 160), 7, 0, [0, 0], 1.0), ((64, 72, 136), 11, 2, [0, 0], 1.0),
                                                      ((72, 64, 232), 4, 0, [0, 0, 0], 1.0), ((68, 52, 272), 6, 1, [0, 0, 0, 0], 0.5),
                                                      ((510, 37, 130), 5, 0, [0, 0], 1.0)])
def test_c_zslab_driver_matches_single_gpu(built, dims, seed, mode, devs, scale):
    vol = vol_of(built, dims, seed)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract(initial_image_scale=scale, desc_mode=mode, size_factor=scale)
    got, st = built.extract_zslab(vol, devs, initial_image_scale=scale, desc_mode=mode, size_factor=scale)
    assert len(want) > 50 and len(got) == len(want)
    assert got.tobytes() == want.tobytes()                                  # bit-identical records, same order
    assert st["n_ranks"] == len(devs) and st["sharded_octaves"] >= 1 and st["n_records"] == len(want)
    # per interface and sharded octave, both directions: 8 + 11 + 15 + 12 slices deferred (round 5: the subsample's eight slices of
    # L3, then what patches of L1 / L2 / L3 reach -- 19 / 23 / 28 slices, not 32); on the critical path the 8-slice halos of
    # L1..L3 and 9 slices of L4 (the 17-tap level is only evaluated around candidates, from L4) -- or, for rows that are not
    # whole 16-byte vectors, five 8-slice halos (every level stored)
    # (a sharded octave whose rows are not whole vectors -- 68 -> 34 -- stores every level while the octave above it does not)
    d, c = st["halo_bytes_deferred"], st["halo_bytes_critical"]
    assert c > 0 and 33 * d <= 46 * c <= 40 * d
    if all((dims[0] >> o) % 4 == 0 for o in range(st["sharded_octaves"])):
        assert 33 * d == 46 * c
        assert 46 * st["halo_bytes_subsample"] == 8 * d               # all the next octave waits for of the deferred slices
        assert st["halo_bytes_hidden"] == c                              # bands first: all of it travels beside the interior launch
    assert st["gather_bytes"] > 0
    # the round-2 schedule (a level in one piece, then its exchange) gives the same bytes, with nothing hidden
    with built.ZSlab(dims[0], dims[1], dims[2], devs) as h:
        h.set_tuning(built.TUNE_BANDS_FIRST, 0)
        got2, st2 = h.extract(vol, initial_image_scale=scale, desc_mode=mode, size_factor=scale)
    assert got2.tobytes() == want.tobytes() and st2["halo_bytes_hidden"] == 0 and st2["halo_bytes_critical"] == c


@pytest.mark.parametrize("dims,seed,devs,mode,noise", [((512, 512, 512), 20240607, [0] * 8, 0, 0.0), ((256, 256, 256), 3, [0] * 4, 2, 0.0),
                                                       ((160, 144, 384), 9, [0] * 6, 0, 12.0), ((96, 80, 160), 7, [0, 0], 3, 0.0)])
def test_c_zslab_patches_stay_within_the_fetched_halos(built, dims, seed, devs, mode, noise):
    """Round 5 fetches of L1 / L2 / L3 only the 19 / 23 / 28 slices beyond a slab that a patch of that level can reach (the buffers
    keep 32).  With SIFT3D_ZSLAB_POISON_HALO every slice that is NOT fetched holds NaN: one sample beyond the bound and the record's
    descriptor, orientation or eigenvalues would differ from the single-device bytes.  """
    vol = vol_of(built, dims, seed)
    if noise:
        rng = np.random.default_rng(seed)
        vol = (vol + rng.normal(0.0, noise, vol.shape)).astype(np.float32)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract(desc_mode=mode)
    with built.ZSlab(dims[0], dims[1], dims[2], devs) as h:
        h.set_tuning(built.ZSLAB_POISON_HALO, 1)
        got, st = h.extract(vol, desc_mode=mode)
        assert st["sharded_octaves"] >= 1 and len(want) > 100
        assert got.tobytes() == want.tobytes()
        h.set_tuning(built.TUNE_BANDS_FIRST, 0)
        h.set_tuning(built.TUNE_LAZY_LEVELS, 0)
        got, st = h.extract(vol, desc_mode=mode)
        assert got.tobytes() == want.tobytes()
        if dims[2] >= 384:
            # the check can fail: with the NaN started k slices INSIDE what was fetched some k changes the records -- the margin
            # the bound has on this volume (at least the one slice of slack it was given), printed for DESIGN.md
            h.set_tuning(built.TUNE_BANDS_FIRST, 1)
            h.set_tuning(built.TUNE_LAZY_LEVELS, 1)
            first = None
            for k in range(1, 12):
                h.set_tuning(built.ZSLAB_POISON_HALO, 1 + k)
                got, st = h.extract(vol, desc_mode=mode)
                if got.tobytes() != want.tobytes():
                    first = k
                    break
            print("poisoned halos, %s in %d slabs: records change when the NaN start %s slices inside the fetched halos"
                  % (dims, len(devs), first), flush=True)
            assert first is not None and first >= 2
    # the same midpoint argument from below: no scale under sigma_h + sigma_c of the finest level, 1.6 (1 + 2^(1/3)) = 3.616 (the
    # bounds of consecutive levels and octaves tile [3.616, inf), so the records can show this side of the claim only)
    assert np.isfinite(want["scale"]).all() and (want["scale"] >= np.float32(1.6 * (1 + 2 ** (1.0 / 3)) * (1 - 1e-6))).all()


def test_c_zslab_every_octave_sharded(built):
    """A long thin volume whose every octave is sharded (slabs of 1024, 512, 256, 128 slices): no octave is gathered, so there is no rank
    for gathered octaves either -- the branch of the driver in which the slabs' ranks are all there is."""
    dims = (40, 36, 2048)
    vol = vol_of(built, dims, 17)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract()
    for devs in ([0, 0], [0, 0, 0, 0]):
        got, st = built.extract_zslab(vol, devs)
        assert got.tobytes() == want.tobytes() and len(want) > 20, (len(got), len(want))
        if len(devs) == 2:
            assert st["sharded_octaves"] == 4 and st["gather_bytes"] == 0 and st["list_grown"] == 0


def test_c_zslab_coarse_octaves_append_behind_the_slabs(built):
    """The octaves that are not sharded have a rank of their own whose records are appended behind the slabs' (round 5): with no room
    left behind them (SIFT3D_ZSLAB_LIST_ROOM 0) the list is replaced by a larger one after the slabs' kernels have stored into it --
    the same bytes, and the stats say that it happened; with the default room it does not happen."""
    dims = (192, 160, 256)
    vol = vol_of(built, dims, 7)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract()
    with built.ZSlab(dims[0], dims[1], dims[2], [0] * 4) as h:
        got, st = h.extract(vol)
        assert got.tobytes() == want.tobytes() and st["list_grown"] == 0
        K = st["sharded_octaves"]
        n_coarse = int((want["scale"] >= 3.6 * 2 ** K).sum())     # records of octave o have scales in [3.616, 7.232] * 2^o
        assert 1 <= K <= 3 and 0 < n_coarse < len(want) // 8
        h.set_tuning(built.ZSLAB_LIST_ROOM, 0)
        got, st = h.extract(vol)
        assert got.tobytes() == want.tobytes() and st["list_grown"] == 1
        got, st = h.extract(vol)                      # the larger list is kept
        assert got.tobytes() == want.tobytes() and st["list_grown"] == 0
        h.set_volume(vol)
        h.set_tuning(built.ZSLAB_LIST_ROOM, 0)
        got, st = h.extract_resident()
        assert got.tobytes() == want.tobytes() and st["list_grown"] == 1
        h.set_tuning(built.ZSLAB_LIST_ROOM, n_coarse)   # exactly enough
        got, st = h.extract_resident()
        assert got.tobytes() == want.tobytes() and st["list_grown"] == 0


def test_c_zslab_handle_is_reusable(built):
    """sift3d_zslab_create / _extract / _destroy: several volumes of one shape through one handle (the first run sizes the
    arena the later ones use), alternating descriptor modes; every result is the single-GPU one, byte for byte."""
    dims = (80, 72, 200)
    vols = [vol_of(built, dims, sd) for sd in (41, 42)] + [np.zeros(dims[::-1], np.float32)]
    with built.Context(*dims) as ctx:
        want = {}
        for i, v in enumerate(vols):
            ctx.set_volume(v)
            for mode in (0, 3):
                want[(i, mode)] = ctx.extract(desc_mode=mode)
    with built.ZSlab(dims[0], dims[1], dims[2], [0, 0, 0]) as h:
        for rnd in range(2):
            for i, v in enumerate(vols):
                for mode in (0, 3):
                    got, st = h.extract(v, desc_mode=mode)
                    assert got.tobytes() == want[(i, mode)].tobytes(), (rnd, i, mode)
                    assert st["n_ranks"] == 3 and st["wall_ms"] > 0
    assert len(want[(0, 0)]) > 100 and len(want[(2, 0)]) == 0
    with pytest.raises(built.Sift3DError):
        built.ZSlab(dims[0], dims[1], dims[2], [0, 42])


def test_config_c5_shape_of_work_on_one_gpu(built):
    """BASELINE config C5 (2048 x 2048 x 1024 over 8 GPUs, NRRIEF) needs eight GPUs at its own size.  Its shape of work --
    eight Z-slabs of 128 slices, six ranks with a neighbour on both sides, three sharded octaves, the NRRIEF descriptor -- is
    rehearsed here with the rows and columns cut to 192 x 160: the C driver with device 0 listed eight times against the
    single-GPU extraction of the same volume, byte for byte."""
    dims = (192, 160, 1024)
    vol = vol_of(built, dims, 2025)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract(desc_mode=built.DESC_NRRIEF)
    got, st = built.extract_zslab(vol, [0] * 8, desc_mode=built.DESC_NRRIEF)
    assert st["n_ranks"] == 8 and st["sharded_octaves"] == 3
    assert len(want) > 5000 and got.tobytes() == want.tobytes()
    assert st["halo_bytes_deferred"] * 33 == st["halo_bytes_critical"] * 46   # 8 + 8 + 8 + 9 slices of L1..L4 per level
    assert st["halo_bytes_hidden"] == st["halo_bytes_critical"]              # all of them issued bands-first


def test_config_c5_plane_size_on_one_gpu(built):
    """BASELINE config C5's own plane size: 2048 x 2048 rows and columns (2^22 voxels, 16 MiB per plane), NRRIEF.  At its
    full depth (1024 slices, 8 GPUs) a rank holds 128 slices and its halos; here two such ranks -- 2048 x 2048 x 256, 2^30
    voxels, three sharded octaves as in C5 -- run through the C slab driver on device 0 and must give the bytes of the
    single-context extraction of the same volume.  This is where a 32-bit plane or byte offset in the kernels would show:
    a slab buffer is 2.7 GB, the volume 4.3 GB, and level offsets pass 2^32 bytes."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * 2 ** 30:
        pytest.skip("needs about 160 GB of free HBM")
    dims = (2048, 2048, 256)
    vol = vol_of(built, dims, 2026)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract(desc_mode=built.DESC_NRRIEF).copy()
        t = ctx.timings()
    assert t["n_octaves"] == 7 and len(want) > 500000
    _record_properties(want, dims, rank_desc=False)
    # keypoints in the far corner of the volume exist (indices beyond 2^31 bytes into a level were addressed)
    assert ((want["z"] > 200) & (want["y"] > 1800) & (want["x"] > 1800)).any()
    got, st = built.extract_zslab(vol, [0, 0], desc_mode=built.DESC_NRRIEF)
    assert st["n_ranks"] == 2 and st["sharded_octaves"] == 3
    assert len(got) == len(want) and got.tobytes() == want.tobytes()
    assert st["halo_bytes_deferred"] * 33 == st["halo_bytes_critical"] * 46
    assert st["halo_bytes_hidden"] == st["halo_bytes_critical"]
    del got
    orc = _oracle_on_all_cores("properties and the two-slab run against the single context")   # last: may skip
    cpu, _ = orc.extract(vol, desc_mode=3)                                  # the CPU restatement agrees, record by record
    assert _compare_records(want, cpu), "float fields are within 1e-4 but not bit-identical"


def test_config_c5_at_its_own_size(built):
    """BASELINE config C5 ITSELF: 2048 x 2048 x 1024 = 2^32 voxels, NRRIEF -- a size the reference cannot address at all (its
    element count is an unsigned int: R/src_common/FeatureIO.cpp:381; its CUDA path indexes with int:
    R/cuda_common/SIFT_cuda_Tools.cu:187).  On ONE 288 GB GPU: (1) the single-context extraction (a pyramid of 11.4 floats a
    voxel, 208 GB -- the pass intermediates of the three-launch blur are allocated on demand above 2^31 voxels);
    (2) properties no oracle is needed for: record invariants, idempotence, the reference's raster order of the candidates,
    keypoints whose linear index is next to 2^32; (3) C5's slab geometry -- Z-slabs of 128 slices of 2048 x 2048, three sharded
    octaves -- through the C slab driver: FOUR of them (2048 x 2048 x 512: eight would need 310 GB on the one device), byte for
    byte the single-context records of that half;
    (4) the translation property of tools/big_volume_check.py at this size: a blob block in the far corner gives the
    candidates of the same block in a small volume, shifted, with bit-identical DoG values (that small volume is one the
    oracle-checked tests cover).  The CPU restatement itself would need 160 GB of host memory and minutes here."""
    import time
    import psutil
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 245 * 2 ** 30:
        pytest.skip("C5 at its own size NOT run: needs about 235 GB of free HBM, this device has %.0f" % (free / 2 ** 30))
    avail = psutil.virtual_memory().available
    if avail < 48 * 2 ** 30:
        pytest.skip("C5 at its own size NOT run: needs about 40 GB of free host memory (the 17 GB volume, its records), found %.0f" % (avail / 2 ** 30))
    dims = (2048, 2048, 1024)
    t0 = time.time()
    vol = vol_of(built, dims, 2027)
    print("C5: volume generated in %.1f s" % (time.time() - t0), flush=True)
    with built.Context(*dims) as ctx:
        t0 = time.time()
        ctx.set_volume(vol)
        want = ctx.extract(desc_mode=built.DESC_NRRIEF)
        t = ctx.timings()
        print("C5: single context: upload + first extraction %.1f s, %d records, %d extrema, %d octaves" %
              (time.time() - t0, len(want), t["n_extrema"], t["n_octaves"]), flush=True)
        t0 = time.time()
        again = ctx.extract(desc_mode=built.DESC_NRRIEF, copy=False)
        print("C5: second extraction (resident volume) %.3f s" % (time.time() - t0), flush=True)
        assert len(again) == len(want) and (again.view(np.uint8) == want.view(np.uint8)).all()      # idempotent
        cands = ctx.detect()
    assert t["n_octaves"] == 9 and len(want) > 4000000 and len(cands) == t["n_extrema"] > 1000000
    _record_properties(want, dims, rank_desc=False)
    key = (cands["octave"].astype(np.int64) << 44) + (cands["level"].astype(np.int64) << 42) + (cands["is_max"].astype(np.int64) << 41)
    lin = ((cands["z"].astype(np.int64) * (dims[1] >> cands["octave"]) + cands["y"]) * (dims[0] >> cands["octave"])) + cands["x"]
    assert lin.max() > 2 ** 32 - 2 ** 27                                        # candidates at the far end of the 2^32 indices
    order = key + lin
    assert (order[1:] > order[:-1]).all()                                       # octave, level, minima first, raster z-y-x: the reference's order
    assert ((want["z"] > 1000) & (want["y"] > 2000) & (want["x"] > 2000)).any()  # keypoints in the far corner
    del cands, again
    del want
    # C5's own eight slabs keep 128 + 2 x 32 slices of every level each: 1.5 x the pyramid, 310 GB -- more than the ONE device
    # of this box holds (on eight devices: 39 GB each).  Half the volume in FOUR slabs of the same 128 slices (two interior
    # ranks, three sharded octaves) is what fits: byte for byte the single-context records of that half.
    half = vol[:512]
    with built.Context(2048, 2048, 512) as ctx:
        ctx.set_volume(half)
        want = ctx.extract(desc_mode=built.DESC_NRRIEF)
    t0 = time.time()
    got, st = built.extract_zslab(half, [0] * 4, desc_mode=built.DESC_NRRIEF)
    print("C5: 2048 x 2048 x 512 in four Z-slabs of 128 slices on one device: %.1f s, %d records" % (time.time() - t0, len(got)), flush=True)
    assert st["n_ranks"] == 4 and st["sharded_octaves"] == 3
    assert len(got) == len(want) > 2000000 and got.tobytes() == want.tobytes()
    assert st["halo_bytes_deferred"] * 33 == st["halo_bytes_critical"] * 46
    assert st["halo_bytes_hidden"] == st["halo_bytes_critical"]
    del got, want, vol, half
    t0 = time.time()
    assert _tool("big_volume_check").check(dims)
    print("C5: translation property at 2048 x 2048 x 1024: %.1f s" % (time.time() - t0), flush=True)


def test_c_zslab_driver_edge_cases(built):
    vol = vol_of(built, (48, 40, 60), 3)                                    # too thin for 32-slice slabs: one rank, the serial path
    with built.Context(48, 40, 60) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract()
    got, st = built.extract_zslab(vol, [0, 0])
    assert got.tobytes() == want.tobytes() and st["n_ranks"] == 1 and st["sharded_octaves"] == 0 and st["halo_bytes_critical"] == 0
    got, st = built.extract_zslab(vol, [0])
    assert got.tobytes() == want.tobytes()
    zero, st = built.extract_zslab(np.zeros((160, 40, 36), np.float32), [0, 0])   # nothing to find, sharded
    assert len(zero) == 0 and st["sharded_octaves"] >= 1
    with pytest.raises(built.Sift3DError) as ei:
        built.extract_zslab(vol, [0, 99])
    assert ei.value.code == -1 and "no HIP device 99" in str(ei.value)


def test_c_zslab_driver_rccl_transport_selection(built):
    """RCCL as the slab driver's transport (round-3 review, item 7; BASELINE north star: "halo exchange over RCCL/xGMI").  What
    one GPU can show: (1) a device listed twice cannot be two RCCL ranks -- the driver falls back to peer copies, says so in
    its stats and returns the single-GPU bytes; (2) with one distinct device the library IS loaded at run time and both
    communicator sets are created and destroyed (ncclCommInitAll over one device; nothing to exchange); (3) a library that
    cannot be loaded is SIFT3D_ERR_COMM with the reason, and leaves the peer-copy transport usable.  Sends and receives between
    two GPUs have never run."""
    dims = (96, 80, 160)
    vol = vol_of(built, dims, 7)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract()
    got, st = built.extract_zslab(vol, [0, 0], transport=built.TRANSPORT_RCCL)
    assert got.tobytes() == want.tobytes() and st["n_ranks"] == 2
    assert st["transport"] == built.TRANSPORT_PEER_COPY and st["transport_fell_back"] == 1 and st["rccl_version"] == 0
    got, st = built.extract_zslab(vol, [0, 0])
    assert st["transport"] == built.TRANSPORT_PEER_COPY and st["transport_fell_back"] == 0 and got.tobytes() == want.tobytes()
    got, st = built.extract_zslab(vol, [0], transport=built.TRANSPORT_RCCL)
    assert got.tobytes() == want.tobytes() and st["n_ranks"] == 1
    assert st["transport"] == built.TRANSPORT_RCCL and st["transport_fell_back"] == 0 and st["rccl_version"] >= 20000
    with built.ZSlab(dims[0], dims[1], dims[2], [0, 0]) as h:           # through the handle: chosen, changed, chosen again
        h.set_tuning(built.ZSLAB_TRANSPORT, built.TRANSPORT_RCCL)
        got, st = h.extract(vol)
        assert got.tobytes() == want.tobytes() and st["transport_fell_back"] == 1
        h.set_tuning(built.ZSLAB_TRANSPORT, built.TRANSPORT_PEER_COPY)
        got, st = h.extract(vol)
        assert got.tobytes() == want.tobytes() and st["transport_fell_back"] == 0
        with pytest.raises(built.Sift3DError):
            h.set_tuning(built.ZSLAB_TRANSPORT, 7)
    built.zslab_set_transport_library("/nonexistent/librccl-none.so")
    try:
        with pytest.raises(built.Sift3DError) as ei:
            built.extract_zslab(vol, [0], transport=built.TRANSPORT_RCCL)
        assert ei.value.code == -5 and "cannot load the RCCL library /nonexistent/librccl-none.so" in str(ei.value)
        got, st = built.extract_zslab(vol, [0])                            # peer copies need no library
        assert got.tobytes() == want.tobytes()
    finally:
        built.zslab_set_transport_library(None)
    got, st = built.extract_zslab(vol, [0], transport=built.TRANSPORT_RCCL)
    assert st["transport"] == built.TRANSPORT_RCCL and got.tobytes() == want.tobytes()


def test_cli_several_devices(built, tmp_path):
    """featExtract -d0,0 (one Z-slab per listed device) writes the .key of featExtract -d0, with and without -2+."""
    nii = str(tmp_path / "in.nii")
    built.write_nifti(nii, vol_of(built, (72, 64, 168), 21), voxel=(1.0, 1.0, 2.0))
    small = str(tmp_path / "s.nii")
    built.write_nifti(small, vol_of(built, (40, 36, 84), 22))
    for flags, src in (([], nii), (["-br"], nii), (["-2+", "-b"], small)):
        k1, k2 = str(tmp_path / "one.key"), str(tmp_path / "two.key")
        r1 = subprocess.run([built.FEATEXTRACT, "-d0"] + flags + [src, k1], capture_output=True, text=True)
        r2 = subprocess.run([built.FEATEXTRACT, "-d0,0"] + flags + [src, k2], capture_output=True, text=True,
                            env=dict(os.environ, SIFT3D_CLI_TIMES="1"))
        assert r1.returncode == 0 and r2.returncode == 0, r2.stdout + r2.stderr
        # the same lines but for the pyramid's debug output ('#<microseconds>' / 'done.' per octave), which only the
        # single-device run has a pyramid of its own to print for
        core = lambda out: [l for l in out.split("\n") if l and not l.startswith("#") and l != "done."]
        assert core(r1.stdout) == core(r2.stdout) and r2.stdout.endswith("\nDone.\n")
        assert "# z-slabs: 2 ranks" in r2.stderr
        assert open(k1, "rb").read() == open(k2, "rb").read() and len(open(k1).readlines()) > 20
    r = subprocess.run([built.FEATEXTRACT, "-d0,7", nii, str(tmp_path / "x.key")], capture_output=True, text=True)
    assert r.returncode == 255 and "Error: unknown device: 7" in r.stdout
    # the transport by name: -d0,0:rccl (one device twice: falls back to peer copies and says so), unknown names refused
    k3 = str(tmp_path / "three.key")
    r3 = subprocess.run([built.FEATEXTRACT, "-d0,0:rccl", nii, k3], capture_output=True, text=True, env=dict(os.environ, SIFT3D_CLI_TIMES="1"))
    assert r3.returncode == 0 and "over peer copies (RCCL asked for, but a device is listed twice)" in r3.stderr
    k1 = str(tmp_path / "one.key")
    subprocess.run([built.FEATEXTRACT, "-d0", nii, k1], check=True, capture_output=True)
    assert open(k3, "rb").read() == open(k1, "rb").read()
    r = subprocess.run([built.FEATEXTRACT, "-d0,0:smoke", nii, k3], capture_output=True, text=True)
    assert r.returncode == 255 and "Error: unknown slab transport: smoke" in r.stdout


# ---------------------------------------------------------------------------------------------------
# Z-slab mode on the GPU: two processes share the one GPU of the test box and exchange halos through gloo
# (staged through the host); on a multi-GPU node the same driver runs with backend "nccl" (RCCL).
# ---------------------------------------------------------------------------------------------------
def _zslab_worker(rank, world, port, dims, seed, mode, q):
    import importlib
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    pkg = importlib.import_module("3d_sift_cuda_amd")
    zs = importlib.import_module("3d_sift_cuda_amd.zslab")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        vol = pkg.synth_blobs(*dims, seed=seed)
        plan = zs.SlabPlan(dims[0], dims[1], dims[2], world)
        i0, i1 = plan.input_range(rank)
        ctx = pkg.Context(dims[0], dims[1], max(zs.slab_context_slices(plan, rank), 8 + 2 * zs.HALO), device=0, slab=True)
        be = zs.HipBackend(pkg, ctx, torch)
        dgroup = dist.new_group(ranks=list(range(world)), backend="gloo")   # the deferred patch halos on a group of their own
        # rank 0: the octaves that are not sharded on a second context and stream, queued by a second host thread (round 5)
        cdims = zs.coarse_octave_dims(plan)
        cctx = pkg.Context(cdims[0], cdims[1], cdims[2], device=0, slab=True) if (rank == 0 and cdims) else None
        cbe = zs.HipBackend(pkg, cctx, torch) if cctx is not None else None
        with be.stream_scope():
            # poison_halo: the halo slices of L1..L3 that the exchange does not fetch hold NaN (zslab.PATCH_REACH is the claim)
            ex = zs.ZSlabExtractor(be, plan, rank, dist, deferred_group=dgroup, poison_halo=True, coarse_backend=cbe)
            ex.run(vol[i0:i1], i0)
            recs, grp = ex.describe(desc_mode=mode)
            stats = dict(ex.stats)
            gathered = [None] * world
            dist.all_gather_object(gathered, (recs, grp))
            merged = zs.merge_by_group(gathered) if rank == 0 else None
            # round 5: the same volume once more, the records stored by every rank's descriptor kernel at their places in ONE list in
            # shared memory (no gather, no merge) -- and once into a list that is too small, which must fall back to each rank's own buffers
            total = [len(merged) if rank == 0 else 0]
            dist.broadcast_object_list(total, src=0)
            placed, small = None, None
            if dims[0] * dims[1] * dims[2] <= 2 ** 28:
                shared = zs.SharedRecordList(pkg, dist, rank, total[0] + 100, pkg.FEATURE_DTYPE)
                ex.run(vol[i0:i1], i0)
                n, own = ex.describe_into(shared, desc_mode=mode, device="cuda:0")
                assert n == total[0] and own is None, (n, total)
                placed = shared.view(n).copy() if rank == 0 else None
                tiny = zs.SharedRecordList(pkg, dist, rank, 10, pkg.FEATURE_DTYPE)
                ex.run(vol[i0:i1], i0)
                n, own = ex.describe_into(tiny, desc_mode=mode, device="cuda:0")
                assert n is None and own is not None
                small = zs.gather_records(dist, rank, world, own[0], own[1], "cuda:0", dtype=pkg.FEATURE_DTYPE)
                tiny.close()
                # one record too few: the slabs' records fit and the coarser octaves' (appended by rank 0) do not -- rank 0 then keeps a
                # list of its own for the run; or, if those octaves have no record here, the slabs' do not fit and everything is gathered
                tight = zs.SharedRecordList(pkg, dist, rank, total[0] - 1, pkg.FEATURE_DTYPE)
                ex.run(vol[i0:i1], i0)
                n, own = ex.describe_into(tight, desc_mode=mode, device="cuda:0")
                if n is None:
                    close = zs.gather_records(dist, rank, world, own[0], own[1], "cuda:0", dtype=pkg.FEATURE_DTYPE)
                else:
                    assert n == total[0] and (rank != 0 or tight.overflow is not None)
                    close = tight.view(n).copy() if rank == 0 else None
                tight.close()
                shared.close()
                if rank == 0:
                    small = (small, close)
                # a rank that cannot register the segment: EVERY rank's constructor raises (nobody waits in a collective for it)
                orig = pkg.host_register
                if rank == world - 1:
                    def refuse(address, nbytes):
                        raise pkg.Sift3DError("injected: no registration on this rank")
                    pkg.host_register = refuse
                try:
                    zs.SharedRecordList(pkg, dist, rank, 1000, pkg.FEATURE_DTYPE)
                    raised = False
                except RuntimeError:
                    raised = True
                finally:
                    pkg.host_register = orig
                assert raised, rank
        if rank == 0:
            q.put((plan.n_sharded, merged, stats, placed, small))
        if cctx is not None:
            cctx.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,seed,mode,world", [((96, 80, 160), 7, 0, 2), ((64, 72, 136), 11, 2, 2), ((72, 64, 232), 4, 0, 3),
                                                  ((1024, 1024, 512), 12345, 0, 4)])
def test_zslab_processes_match_single_gpu(built, dims, seed, mode, world):
    """Two ranks, and three (the middle one exchanges on both sides, as every interior rank of an 8-GPU run does); the last
    case is BASELINE config C4 itself -- the 1024 x 1024 x 512 volume cut into its four Z-slabs -- rehearsed with four
    processes on the one GPU of the box (halos staged through the host over gloo; on a 4-GPU node the same driver runs
    over RCCL)."""
    if dims[0] * dims[1] * dims[2] > 2 ** 28:
        import torch
        free, _ = torch.cuda.mem_get_info()
        if free < 120 * 2 ** 30:
            pytest.skip("needs about 90 GB of free HBM (four slab contexts, then the single-GPU context)")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_zslab_worker, args=(r, world, port, dims, seed, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    n_sharded, merged, stats, placed, small = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # per sharded octave: the 8-slice halos of L1..L3 and 9 slices of L4 on the critical path (the 17-tap level is only evaluated
    # around candidates, from L4), two deferred batches (8 slices of L3 for the subsample; 11 + 15 + 12 slices that only patches reach); an octave whose rows are not whole 16-byte vectors stores
    # every level: five 8-slice halos
    lazy = [((dims[0] >> o) % 4 == 0 and (dims[0] >> o) >= 8) for o in range(n_sharded)]
    assert n_sharded >= 1 and stats["deferred_exchanges"] == 2 * n_sharded
    assert stats["exchanges"] == sum(6 if z else 7 for z in lazy)
    if all(lazy):
        assert stats["deferred_bytes"] * 79 == stats["exchange_bytes"] * 46
        # boundary bands first: every per-level halo was issued behind the band launches and moved beside the interior launch
        assert stats["hidden_bytes"] == stats["exchange_bytes"] - stats["deferred_bytes"]
    elif not any(lazy):
        assert stats["deferred_bytes"] * 86 == stats["exchange_bytes"] * 46
        assert stats["hidden_bytes"] == 0                                # no windowed blur for such rows: level, then exchange
    vol = vol_of(built, dims, seed)
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        want = ctx.extract(desc_mode=mode)
    assert len(want) > 50 and len(merged) == len(want)
    assert (merged.view(np.uint8) == want.view(np.uint8)).all()   # bit-identical records, same order
    if placed is not None:   # every rank's kernel stored its records in the one shared list / the list was too small: same bytes
        assert placed.tobytes() == want.tobytes() and small[0].tobytes() == want.tobytes() and small[1].tobytes() == want.tobytes()
    else:
        assert dims[0] * dims[1] * dims[2] > 2 ** 28


def test_roofline_ceiling_probe_runs(built):
    """tools/_build/libroof.so (bench.py's roofline.ceiling: the zero-arithmetic march of the fused blur's tiles): four march times
    that are positive, ordered as their byte counts, and faster than any blur of the same volume could be."""
    import ctypes
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_build", "libroof.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(lib))], check=True, capture_output=True)
    L = ctypes.CDLL(lib)
    L.roof_march.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    ms = (ctypes.c_float * 4)()
    assert L.roof_march(256, 5, ms) == 0
    one, two = min(ms[0], ms[1]), min(ms[2], ms[3])
    assert 0.0 < one < two < 1.0, list(ms)                      # 256^3: 16.8 M voxels, 8 and 12 B each: tens of microseconds
    assert 8.0 * 256 ** 3 / (one * 1e-3) / 1e9 < 8000.0          # below the HBM peak
    assert L.roof_march(100, 5, ms) != 0                         # a volume its tiles do not divide is refused, not marched out of bounds
