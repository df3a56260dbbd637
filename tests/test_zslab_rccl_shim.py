"""GPU suite: the RCCL half of the C slab driver's exchange, rehearsed on ONE GPU (round-4 review item 1c, advisor finding).

Real RCCL refuses two ranks on one device, and the development box has one, so `zs_xfer`'s RCCL branch
(`3d_sift_cuda_amd/csrc/zslab_transport.hip`: ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd, two communicator
sets, the streams the operations are ordered in) had never executed.  Here the driver is pointed at the rehearsal library
`tests/rccl_shim` (TEST infrastructure: eight `nccl*` symbols that pair the sends and receives of a group, check them, and
run them as stream-ordered copies with RCCL's rendezvous semantics) and told to keep duplicate devices as RCCL ranks
(`SIFT3D_ZSLAB_DUPLICATE_RANKS`).  Asserted: the merged records are the single-GPU bytes for 2, 3, 4 and 8 ranks; every send
met exactly one receive of the same size in its group; nothing was posted outside a group; a communicator saw one stream per
group; both communicator sets carried traffic (one with `SIFT3D_ZSLAB_SERIAL_CHANNELS`); the bytes the library moved are the
bytes the driver's stats claim; repeated runs through one handle stay identical.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "rccl_shim", "_build", "librccl_shim.so")
STAT_NAMES = ["groups", "sends", "recvs", "pairs", "bytes", "unmatched", "ungrouped", "count_mismatch", "self_sends", "multi_stream",
              "max_ops_per_group", "comm_sets", "destroyed_in_group", "hip_errors", "open_depth", "live_comms", "sets_with_traffic",
              "bytes_set_a", "bytes_set_b", "dead_comm"]


@pytest.fixture(scope="module")
def shim(built):
    if not os.path.exists(SHIM):
        subprocess.run(["make", "-C", os.path.join(HERE, "rccl_shim")], check=True, capture_output=True)
    lib = C.CDLL(SHIM)   # the same mapping the driver's dlopen returns: one set of counters
    lib.rccl_shim_stats.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.rccl_shim_last_message.restype = C.c_char_p

    class Shim:
        def reset(self):
            lib.rccl_shim_reset()

        def stats(self):
            buf = (C.c_int64 * len(STAT_NAMES))()
            n = lib.rccl_shim_stats(buf, len(STAT_NAMES))
            assert n == len(STAT_NAMES)
            return dict(zip(STAT_NAMES, list(buf)))

        def message(self):
            return lib.rccl_shim_last_message().decode()

    built.zslab_set_transport_library(SHIM)
    yield Shim()
    built.zslab_set_transport_library(None)


def clean(s):
    """what every run must leave in the library's counters"""
    for k in ("unmatched", "ungrouped", "count_mismatch", "self_sends", "multi_stream", "destroyed_in_group", "hip_errors", "open_depth",
              "dead_comm"):
        assert s[k] == 0, (k, s)
    assert s["sends"] == s["recvs"] == s["pairs"], s


def single_gpu(built, dims, vol, **kw):
    with built.Context(*dims) as ctx:
        ctx.set_volume(vol)
        return ctx.extract(**kw)


@pytest.mark.parametrize("dims,ranks,mode", [((96, 80, 160), 2, 0), ((72, 64, 200), 3, 2), ((64, 48, 256), 4, 0), ((48, 40, 512), 8, 3),
                                             ((70, 52, 160), 2, 1)])
def test_rccl_branch_gives_the_single_gpu_bytes(built, shim, dims, ranks, mode):
    vol = built.synth_blobs(*dims, seed=31 + ranks)
    want = single_gpu(built, dims, vol, desc_mode=mode)
    shim.reset()
    with built.ZSlab(dims[0], dims[1], dims[2], [0] * ranks) as h:
        h.set_tuning(built.ZSLAB_TRANSPORT, built.TRANSPORT_RCCL)
        h.set_tuning(built.ZSLAB_DUPLICATE_RANKS, 1)
        h.set_tuning(built.ZSLAB_POISON_HALO, 1)   # what the exchange does not fetch of L1..L3 holds NaN
        got, st = h.extract(vol, desc_mode=mode)
        s = shim.stats()
        assert st["n_ranks"] == ranks and st["sharded_octaves"] >= 1
        assert st["transport"] == built.TRANSPORT_RCCL and st["transport_fell_back"] == 0 and st["rccl_version"] == -5 and st["comm_sets"] == 2
        assert got.tobytes() == want.tobytes() and len(got) > 20
        clean(s)
        # every byte the driver's stats claim went through the library, and nothing else did
        assert s["bytes"] == st["halo_bytes_critical"] + st["halo_bytes_deferred"] + st["gather_bytes"], (s, st)
        assert s["comm_sets"] == 2 and s["live_comms"] == 2 * ranks
        if st["halo_bytes_deferred"]:
            assert s["sets_with_traffic"] == 2
            assert sorted((s["bytes_set_a"], s["bytes_set_b"])) == sorted((st["halo_bytes_deferred"], st["halo_bytes_critical"] + st["gather_bytes"]))
        # an interior rank exchanges on both sides: one group holds up to 2 sends + 2 receives per rank (6 + 6 in the deferred batch)
        assert s["max_ops_per_group"] >= 2 * (ranks - 1)
        # the resident form takes the same path
        h.set_volume(vol)
        got2, st2 = h.extract_resident(desc_mode=mode)
        assert got2.tobytes() == want.tobytes() and st2["resident_volume"] == 1 and st["resident_volume"] == 0
        clean(shim.stats())
    assert shim.stats()["live_comms"] == 0   # both sets destroyed with the handle


def test_rccl_branch_serial_channels_and_schedules(built, shim):
    """ONE communicator set (SIFT3D_ZSLAB_SERIAL_CHANNELS: the fallback for a first run on real links that stalls with two
    communicators per device), round 2's schedule (no bands first: the halos travel on the main streams), and the stored-level
    schedule: the same bytes each way."""
    dims = (80, 64, 192)
    vol = built.synth_blobs(*dims, seed=5)
    want = single_gpu(built, dims, vol)
    for serial, bands, lazy in ((1, 1, 1), (0, 0, 1), (1, 0, 0), (0, 1, 0)):
        shim.reset()
        with built.ZSlab(dims[0], dims[1], dims[2], [0, 0, 0]) as h:
            h.set_tuning(built.ZSLAB_TRANSPORT, built.TRANSPORT_RCCL)
            h.set_tuning(built.ZSLAB_DUPLICATE_RANKS, 1)
            h.set_tuning(built.ZSLAB_SERIAL_CHANNELS, serial)
            h.set_tuning(built.TUNE_BANDS_FIRST, bands)
            h.set_tuning(built.TUNE_LAZY_LEVELS, lazy)
            got, st = h.extract(vol)
            s = shim.stats()
            assert got.tobytes() == want.tobytes(), (serial, bands, lazy)
            clean(s)
            assert st["comm_sets"] == (1 if serial else 2) and s["comm_sets"] == st["comm_sets"] and s["live_comms"] == 3 * st["comm_sets"]
            assert s["sets_with_traffic"] == st["comm_sets"]
            assert s["bytes"] == st["halo_bytes_critical"] + st["halo_bytes_deferred"] + st["gather_bytes"]
            assert (st["halo_bytes_hidden"] > 0) == bool(bands)
            # the knob may change between two extractions of one handle: the transport is rebuilt
            h.set_tuning(built.ZSLAB_SERIAL_CHANNELS, 1 - serial)
            got, st = h.extract(vol)
            assert got.tobytes() == want.tobytes() and st["comm_sets"] == (2 if serial else 1)
            clean(shim.stats())


def test_rccl_branch_fifty_runs(built, shim):
    """Fifty extractions through one handle (four ranks): a race between the halo streams and the launches that read the halos
    would show as a record that differs from run to run."""
    dims = (64, 56, 256)
    vol = built.synth_blobs(*dims, seed=77)
    want = single_gpu(built, dims, vol, desc_mode=2)
    shim.reset()
    with built.ZSlab(dims[0], dims[1], dims[2], [0, 0, 0, 0]) as h:
        h.set_tuning(built.ZSLAB_TRANSPORT, built.TRANSPORT_RCCL)
        h.set_tuning(built.ZSLAB_DUPLICATE_RANKS, 1)
        h.set_volume(vol)
        first = None
        for i in range(50):
            got, st = h.extract_resident(desc_mode=2, copy=False)
            assert got.tobytes() == want.tobytes(), i
            s = shim.stats()
            clean(s)
            if first is None:
                first = s
        assert s["bytes"] == 50 * first["bytes"] and s["groups"] == 50 * first["groups"] and s["pairs"] == 50 * first["pairs"]


def test_real_rccl_refuses_duplicate_ranks_and_the_handle_recovers(built, shim):
    """With the real library the duplicate list is an error from ncclCommInitAll (SIFT3D_ERR_COMM with its text), not a hang;
    the same handle then runs on peer copies."""
    dims = (96, 80, 160)
    vol = built.synth_blobs(*dims, seed=7)
    want = single_gpu(built, dims, vol)
    built.zslab_set_transport_library(None)
    try:
        with built.ZSlab(dims[0], dims[1], dims[2], [0, 0]) as h:
            h.set_tuning(built.ZSLAB_TRANSPORT, built.TRANSPORT_RCCL)
            h.set_tuning(built.ZSLAB_DUPLICATE_RANKS, 1)
            with pytest.raises(built.Sift3DError) as ei:
                h.extract(vol)
            assert ei.value.code == -5 and "ncclCommInitAll" in str(ei.value)
            h.set_tuning(built.ZSLAB_DUPLICATE_RANKS, 0)   # back to the fall-back: peer copies
            got, st = h.extract(vol)
            assert got.tobytes() == want.tobytes() and st["transport_fell_back"] == 1
    finally:
        built.zslab_set_transport_library(SHIM)


def test_a_failed_exchange_step_leaves_no_group_open(built, shim):
    """Advisor finding (round 4): a transfer that fails between zs_xfer_begin and zs_xfer_end must not leave the RCCL group open
    (the next extraction would nest a second one and the communicators would be destroyed inside it).  A library whose
    ncclSend fails is not at hand, so the path is driven from the other side: the shim refuses operations on communicators it
    has destroyed, which is what the driver's next extraction would use if it kept a transport after a COMM error."""
    dims = (96, 80, 160)
    vol = built.synth_blobs(*dims, seed=7)
    want = single_gpu(built, dims, vol)
    shim.reset()
    with built.ZSlab(dims[0], dims[1], dims[2], [0, 0]) as h:
        h.set_tuning(built.ZSLAB_TRANSPORT, built.TRANSPORT_RCCL)
        h.set_tuning(built.ZSLAB_DUPLICATE_RANKS, 1)
        got, st = h.extract(vol)
        assert got.tobytes() == want.tobytes()
        # destroy the first set's communicators behind the driver's back: its next ncclSend is refused mid-group
        lib = C.CDLL(SHIM)
        lib.rccl_shim_kill_set.argtypes = [C.c_int]
        assert lib.rccl_shim_kill_set(0) == 2
        with pytest.raises(built.Sift3DError) as ei:
            h.extract(vol)
        assert ei.value.code == -5 and "ncclSend / ncclRecv" in str(ei.value)
        s = shim.stats()
        assert s["open_depth"] == 0 and s["destroyed_in_group"] == 0, s     # the group was closed before the transport went
        got, st = h.extract(vol)                                            # fresh communicators
        assert got.tobytes() == want.tobytes() and st["transport"] == built.TRANSPORT_RCCL
        s = shim.stats()
        assert s["open_depth"] == 0 and s["live_comms"] == 4, s
