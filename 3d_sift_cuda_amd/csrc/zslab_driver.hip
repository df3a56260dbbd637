/*
 * zslab_driver.hip -- sift3d_zslab_* / sift3d_extract_zslab*: one volume cut into Z-slabs, ONE process, one context per
 * listed device, built from the api_*.hip per-context building blocks (pipeline.h) and the rank-to-rank transfers of
 * zslab_transport.hip (peer copies or RCCL).  Split from api.hip in round 4 (round-3 review, weak 9).
 */
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pipeline.h"
#include "zs_crew.h"
#include "zslab_transport.h"

/* ======================================================================================================================
 * Z-slab extraction driven from C: ONE process, one context per device, halos moved with hipMemcpyPeerAsync (xGMI between
 * the GPUs of a node).  The reference has no multi-GPU code; the partitioning is the one of
 * 3d_sift_cuda_amd/zslab.py (which runs one process per GPU over RCCL for bench.py): slabs along z with boundaries that are
 * multiples of 2^K, every level recomputed on slab +- 8 slices and its 8-slice halo refreshed from the two neighbours, the
 * rest of the L1..L3 halos copied on a second stream while L4, L5 and the extrema passes run (the eight slices of L3 the
 * subsample reads, then what the patches of each level can reach: ZS_PATCH_REACH), the octaves below the sharded ones gathered
 * on the first device into a rank of their own.  A halo copy is queued on the RECEIVER's stream behind an event the sender
 * records when the level is complete, so no host thread ever waits inside the pyramid; every rank's launches are queued by a
 * host thread of its own (zs_crew.h), step by step of the schedule.  The same device may be listed several times (a rehearsal
 * of the slab logic on one GPU).
 * ====================================================================================================================== */
namespace {
const int64_t ZS_HALO = 32; /* slices of L1..L3 a slab's buffers keep around it (and the least a slab must be thick) */
/* Slices of L_l (l = 1..3) beyond a slab that the patches of its own keypoints can reach (round 5; rounds 1 - 4 fetched ZS_HALO of
 * every level).  A keypoint of detection level l samples L_l at most 5 sqrt(3) patch steps of 2 s / 5 from its centre, s its
 * scale; s = 2 x the vertex of the parabola through (sigma_h, sigma_c, sigma_l) and the three DoG values, and because the centre
 * value is a strict extremum of the three (it passed the tests against both neighbour levels) the vertex lies between the
 * midpoints of the two intervals: s <= sigma_c + sigma_l = 1.6 (2^(l/3) + 2^((l+1)/3)) = 4.556, 5.740, 7.232.  With the
 * centre's own refinement (within [iz, iz + 1]) and the two voxels a trilinear sample reads: ceil(2 sqrt(3) s + 1.5) = 18, 22,
 * 27 slices; one more each for frames that are orthonormal only to rounding.  tests: SIFT3D_ZSLAB_POISON_HALO fills every slice
 * that is NOT fetched with NaN, so a patch that reached further would show in the records. */
const int64_t ZS_PATCH_REACH[4] = {0, 19, 23, 28};
const int64_t ZS_SUB = 16; /* the subsample that seeds the next octave's slab +- 8 reads L3 on slab +- 16 */
const int64_t ZS_BLUR = 8;  /* slices recomputed / exchanged for the next blur (largest filter half-width) */

struct zs_plan {
    int64_t nx, ny, nz;
    int S;
    std::vector<std::vector<int64_t>> oct; /* {X, Y, Z} per octave */
    int K;                                 /* sharded octaves */
    std::vector<int64_t> bounds;           /* S + 1 */
    zs_plan(int64_t nx_, int64_t ny_, int64_t nz_, int S_) : nx(nx_), ny(ny_), nz(nz_), S(S_), K(0)
    {
        int64_t x = nx, y = ny, z = nz;
        while (x > 2 && y > 2 && z > 2 && oct.size() < 32) {
            oct.push_back({x, y, z});
            x /= 2; y /= 2; z /= 2;
        }
        bounds.assign((size_t)S + 1, 0);
        bounds[(size_t)S] = nz;
        if (S <= 1) return;
        /* K = number of sharded octaves: boundaries multiples of 2^K, every slab of octave K-1 at least ZS_HALO thick */
        for (int k = (int)oct.size(); k >= 1; k--) {
            const int64_t align = 1ll << k;
            std::vector<int64_t> b((size_t)S + 1);
            for (int r = 0; r < S; r++) b[(size_t)r] = (int64_t)llround((double)r * (double)nz / S / (double)align) * align;
            b[(size_t)S] = nz;
            bool ok = true;
            for (int r = 0; r < S && ok; r++) ok = b[(size_t)r + 1] > b[(size_t)r];
            for (int o = 0; o < k && ok; o++)
                for (int r = 0; r < S && ok; r++) {
                    const int64_t lo = b[(size_t)r] >> o, hi = r == S - 1 ? oct[(size_t)o][2] : b[(size_t)r + 1] >> o;
                    ok = hi - lo >= ZS_HALO;
                }
            if (ok) {
                K = k;
                bounds = b;
                break;
            }
        }
    }
    void slab(int r, int o, int64_t &z0, int64_t &z1) const
    {
        z0 = bounds[(size_t)r] >> o;
        z1 = r == S - 1 ? oct[(size_t)o][2] : bounds[(size_t)r + 1] >> o;
    }
    void input_range(int r, int64_t &i0, int64_t &i1) const
    {
        int64_t z0, z1;
        slab(r, 0, z0, z1);
        i0 = std::max<int64_t>(0, z0 - 2 * ZS_BLUR);
        i1 = std::min<int64_t>(nz, z1 + 2 * ZS_BLUR);
    }
};

struct zs_rank {
    sift3d_ctx *c = nullptr;
    int dev = 0;
    hipStream_t copy_stream = nullptr; /* the deferred patch-halo copies */
    hipStream_t halo_stream = nullptr; /* the per-level halo copies into this rank, beside its interior launch */
    hipEvent_t ev_halo = nullptr;      /* those copies are done */
    hipEvent_t ev_level = nullptr;     /* this rank's current level is complete (its own slices are final) */
    hipEvent_t ev_l3 = nullptr;        /* L1..L3 of the current octave are complete */
    hipEvent_t ev_patch = nullptr;     /* the slices of L3 the subsample reads beyond +- 8 have arrived (first deferred step) */
    std::vector<float *> allocs;       /* what this run had to allocate beside the arena */
    float *arena = nullptr;            /* one block reused from run to run (sized after the first run of a handle) */
    float *vol_dev = nullptr;          /* sift3d_zslab_set_volume: this rank's input slices [i0, i1), resident between extractions */
    int64_t arena_cap = 0, arena_used = 0, need = 0; /* floats */
    float *L[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, *D[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t z0 = 0, z1 = 0, e0 = 0, e1 = 0; /* current octave: own slices [z0, z1), buffer extent [e0, e1) */
    bool lo = false, hi = false;
    std::vector<sift3d_level> levels;
    float *alloc(int64_t nfloats)
    {
        nfloats = ((nfloats > 0 ? nfloats : 1) + 63) / 64 * 64; /* 256-byte granules */
        need += nfloats;
        if (arena && arena_used + nfloats <= arena_cap) {
            float *p = arena + arena_used;
            arena_used += nfloats;
            return p;
        }
        float *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(float) * (size_t)nfloats) != hipSuccess) return nullptr;
        allocs.push_back(p);
        return p;
    }
    /* end of a run: drop what was allocated beside the arena and make the arena big enough for a run like this one */
    void recycle()
    {
        for (float *p : allocs) hipFree(p);
        allocs.clear();
        if (need > arena_cap) {
            hipFree(arena);
            arena = nullptr;
            arena_cap = 0;
            if (hipMalloc((void **)&arena, sizeof(float) * (size_t)need) == hipSuccess) arena_cap = need;
        }
        arena_used = need = 0;
    }
};


/* Failures inside a rank's step (a lambda run by the rank's host thread, `r` its rank): the failure goes to the rank's own slot and the
 * step returns; the calling thread looks at the slots when every rank is back (crew_failed). */
#define ZR_FAIL(code, ...)                                             \
    do {                                                               \
        rrc[(size_t)r] = (code);                                       \
        snprintf(rerr[(size_t)r].b, sizeof rerr[0].b, __VA_ARGS__);    \
        return;                                                        \
    } while (0)
#define ZR_HIP(call)                                                                                                                      \
    do {                                                                                                                                  \
        hipError_t e_ = (call);                                                                                                           \
        if (e_ != hipSuccess) ZR_FAIL(SIFT3D_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);      \
    } while (0)
#define ZR_COMM(call)                                                                                                                            \
    do {                                                                                                                                         \
        hipError_t e_ = (call);                                                                                                                  \
        if (e_ != hipSuccess) ZR_FAIL(SIFT3D_ERR_COMM, "slab exchange: %s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define ZR_X(call)                                                                     \
    do {                                                                               \
        if ((call) != 0) ZR_FAIL(SIFT3D_ERR_COMM, "%s", zs_transport_error(h->tr));    \
    } while (0)
#define ZR_RC(call)                                                                                         \
    do {                                                                                                    \
        const int rc_ = (call);                                                                             \
        if (rc_ != SIFT3D_OK) ZR_FAIL(rc_, "rank %d: %s", r, sift3d_last_error(R[(size_t)r].c));            \
    } while (0)

struct zs_errline { char b[400]; };
struct zs_tally { int64_t critical = 0, hidden = 0, deferred = 0, subsample = 0, exchanges = 0; char pad[24]; }; /* per rank: a cache line each */
} // namespace

struct sift3d_zslab {
    zs_plan plan;
    std::vector<int> devices;
    std::vector<zs_rank> R; /* one per slab; a single one when the volume is too thin to shard */
    int lazy_levels = 1;    /* SIFT3D_TUNE_LAZY_LEVELS */
    int bands_first = 1;    /* SIFT3D_TUNE_BANDS_FIRST */
    int transport_want = ZS_TRANSPORT_PEER; /* SIFT3D_ZSLAB_TRANSPORT */
    int transport_flags = 0;                /* ZS_FLAG_*: SIFT3D_ZSLAB_SERIAL_CHANNELS, SIFT3D_ZSLAB_DUPLICATE_RANKS */
    int patch_wait = 0;                     /* SIFT3D_ZSLAB_PATCH_WAIT: 0 the patch-only halos are waited for before the per-keypoint stage, 1 at their octave's end */
    int list_room = -1;                     /* SIFT3D_ZSLAB_LIST_ROOM (tests): records of room behind the slabs' in a newly allocated list; -1: an eighth + 4096 */
    int poison_halo = 0;                    /* SIFT3D_ZSLAB_POISON_HALO (tests): the halo slices of L1..L3 that are not fetched hold NaN */
    zs_transport *tr = nullptr;             /* created by the first extraction after the choice (zslab_transport.hip) */
    bool has_volume = false;                /* sift3d_zslab_set_volume has put every rank's input slices on its device */
    /* The merged records (round 5): ONE pinned host buffer every rank's device can store into (hipHostMallocPortable).  A rank's
     * records are sorted by group already, so their merged positions are the rank's own positions shifted group by group: once
     * every rank's records per group are known (a 193-word read-back beside the record total each rank waits for anyway) the
     * descriptor kernels store straight into their merged places and there is no merge left to do. */
    sift3d_feature *merged = nullptr;
    int64_t merged_cap = 0;
    zs_crew crew; /* one host thread per rank beyond the first (started with the handle) */
    int n_slabs = 1; /* R[0 .. n_slabs): the slabs.  R[coarse] (if >= 0): the rank of the gathered octaves, see zslab_create_impl */
    int coarse = -1;
    sift3d_zslab(int64_t nx, int64_t ny, int64_t nz, int n) : plan(nx, ny, nz, n) {}
};

extern "C" void sift3d_zslab_destroy(sift3d_zslab *h)
{
    if (!h) return;
    h->crew.stop();
    for (zs_rank &q : h->R) /* nothing of an exchange may be in flight when its communicators go */
        if (q.c) {
            hipSetDevice(q.dev);
            hipStreamSynchronize(q.c->stream);
            if (q.copy_stream) hipStreamSynchronize(q.copy_stream);
            if (q.halo_stream) hipStreamSynchronize(q.halo_stream);
        }
    zs_transport_destroy(h->tr);
    h->tr = nullptr;
    for (zs_rank &q : h->R) {
        if (!q.c) continue;
        hipSetDevice(q.dev);
        hipStreamSynchronize(q.c->stream);
        if (q.copy_stream) { hipStreamSynchronize(q.copy_stream); hipStreamDestroy(q.copy_stream); }
        if (q.halo_stream) { hipStreamSynchronize(q.halo_stream); hipStreamDestroy(q.halo_stream); }
        if (q.ev_halo) hipEventDestroy(q.ev_halo);
        for (float *p : q.allocs) hipFree(p);
        hipFree(q.arena);
        hipFree(q.vol_dev);
        if (q.ev_level) hipEventDestroy(q.ev_level);
        if (q.ev_l3) hipEventDestroy(q.ev_l3);
        if (q.ev_patch) hipEventDestroy(q.ev_patch);
        sift3d_destroy(q.c);
    }
    if (h->merged) hipHostFree(h->merged);
    delete h;
}

extern "C" int sift3d_zslab_set_tuning(sift3d_zslab *h, int knob, int value)
{
    if (!h) return SIFT3D_ERR_ARG;
    if (knob == SIFT3D_ZSLAB_TRANSPORT) { /* the driver's own: how a block of slices travels between two ranks */
        if (value != SIFT3D_TRANSPORT_PEER_COPY && value != SIFT3D_TRANSPORT_RCCL) return SIFT3D_ERR_ARG;
        if (h->tr && value != h->transport_want) { /* replaced at the next extraction; nothing is in flight between two */
            zs_transport_destroy(h->tr);
            h->tr = nullptr;
        }
        h->transport_want = value;
        return SIFT3D_OK;
    }
    if (knob == SIFT3D_ZSLAB_PATCH_WAIT) {
        if (value != 0 && value != 1) return SIFT3D_ERR_ARG;
        h->patch_wait = value;
        return SIFT3D_OK;
    }
    if (knob == SIFT3D_ZSLAB_LIST_ROOM) {
        h->list_room = value < 0 ? -1 : value;
        if (h->merged) (void)hipHostFree(h->merged); /* the next extraction allocates its list anew */
        h->merged = nullptr;
        h->merged_cap = 0;
        return SIFT3D_OK;
    }
    if (knob == SIFT3D_ZSLAB_POISON_HALO) {
        h->poison_halo = value < 0 ? 0 : value; /* 1 + k: the poison starts k slices INSIDE what was fetched (how much margin the bound has) */
        return SIFT3D_OK;
    }
    if (knob == SIFT3D_ZSLAB_SERIAL_CHANNELS || knob == SIFT3D_ZSLAB_DUPLICATE_RANKS) {
        const int bit = knob == SIFT3D_ZSLAB_SERIAL_CHANNELS ? ZS_FLAG_SERIAL_CHANNELS : ZS_FLAG_DUPLICATE_RANKS;
        const int flags = value ? (h->transport_flags | bit) : (h->transport_flags & ~bit);
        if (h->tr && flags != h->transport_flags) { /* the next extraction builds its transport anew */
            zs_transport_destroy(h->tr);
            h->tr = nullptr;
        }
        h->transport_flags = flags;
        return SIFT3D_OK;
    }
    for (zs_rank &q : h->R) {
        const int rc = sift3d_set_tuning(q.c, knob, value);
        if (rc) return rc;
    }
    if (knob == SIFT3D_TUNE_LAZY_LEVELS) h->lazy_levels = value;
    if (knob == SIFT3D_TUNE_BANDS_FIRST) h->bands_first = value;
    return SIFT3D_OK;
}

/* status_out (may be NULL) receives the sift3d_status behind a NULL result */
static sift3d_zslab *zslab_create_impl(const int *devices, int n_devices, int64_t nx, int64_t ny, int64_t nz, char *err, int64_t err_len,
                                       int *status_out)
{
    char errbuf[512] = "";
    int rc = SIFT3D_OK;
    if (err && err_len > 0) err[0] = 0;
    if (status_out) *status_out = SIFT3D_ERR_ARG;
    auto fail = [&](const char *msg) -> sift3d_zslab * {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", msg);
        return nullptr;
    };
    if (!devices || n_devices < 1 || n_devices > 64 || nx <= 0 || ny <= 0 || nz <= 1) return fail("bad arguments");
    const int ndev = sift3d_device_count();
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= ndev) {
            snprintf(errbuf, sizeof errbuf, "no HIP device %d", devices[i]);
            return fail(errbuf);
        }
    sift3d_zslab *h = new sift3d_zslab(nx, ny, nz, n_devices);
    const int S = h->plan.K > 0 ? n_devices : 1; /* too thin to shard: the whole volume on the first device */
    /* The octaves below the sharded ones are gathered on the first device -- into a rank of their own (round 5): a context, streams
     * and a host thread beside rank 0's, so that the chain of small launches that builds them (0.5 - 0.8 ms at 512^3, latency all
     * of it) and their few hundred keypoints run BESIDE rank 0's per-keypoint stage instead of in front of it.  Every rank's
     * descriptor launch waits for every rank's record counts, so what rank 0 did alone, all ranks waited for. */
    const bool has_coarse = S > 1 && (size_t)h->plan.K < h->plan.oct.size();
    const int T = S + (has_coarse ? 1 : 0);
    h->n_slabs = S;
    h->coarse = has_coarse ? S : -1;
    h->devices.assign(devices, devices + n_devices);
    h->R.resize((size_t)T);
    {
        /* One rank's context, streams and events.  Round 5: the ranks are set up by one host thread each -- a context is 30 - 45 ms of
         * allocations, and on eight devices the one-shot call (featExtract -d0,..,7) spent a third of a second creating them one
         * after the other.  */
        std::vector<int> rcs((size_t)T, SIFT3D_OK);
        std::vector<std::string> errs((size_t)T);
        auto setup = [&](int r) {
            zs_rank &q = h->R[(size_t)r];
            char eb[256];
            auto hipfail = [&](hipError_t e, const char *what) {
                snprintf(eb, sizeof eb, "rank %d: %s failed: %s", r, what, hipGetErrorString(e));
                errs[(size_t)r] = eb;
                rcs[(size_t)r] = SIFT3D_ERR_DEVICE;
            };
            q.dev = r < S ? devices[r] : devices[0];
            int64_t i0 = 0, i1 = nz;
            if (S > 1 && r < S) h->plan.input_range(r, i0, i1);
            hipError_t e = hipSetDevice(q.dev);
            if (e != hipSuccess) return hipfail(e, "hipSetDevice");
            /* a slab context owns no level buffers (they come from the rank's arena); its pass intermediates must hold the
             * largest volume the rank ever blurs: its slab with halos -- or, for the rank of the gathered octaves, the first of them */
            if (r < S) {
                q.c = ctx_create(q.dev, nx, ny, (i1 - i0) + 2 * ZS_HALO, S > 1);
            } else {
                const std::vector<int64_t> &g = h->plan.oct[(size_t)h->plan.K];
                q.c = ctx_create(q.dev, g[0], g[1], g[2], true);
            }
            if (!q.c) {
                snprintf(eb, sizeof eb, "rank %d: no context on device %d (memory?)", r, q.dev);
                errs[(size_t)r] = eb;
                rcs[(size_t)r] = SIFT3D_ERR_MEMORY;
                return;
            }
            if ((e = hipStreamCreateWithFlags(&q.copy_stream, hipStreamNonBlocking)) != hipSuccess) return hipfail(e, "hipStreamCreateWithFlags");
            if ((e = hipStreamCreateWithFlags(&q.halo_stream, hipStreamNonBlocking)) != hipSuccess) return hipfail(e, "hipStreamCreateWithFlags");
            hipEvent_t *evs[] = {&q.ev_halo, &q.ev_level, &q.ev_l3, &q.ev_patch};
            for (hipEvent_t *ev : evs)
                if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return hipfail(e, "hipEventCreateWithFlags");
            for (int p = 0; p < S; p++) /* direct copies between the devices where the fabric allows (errors: already on, or same device) */
                if (devices[p] != q.dev) (void)hipDeviceEnablePeerAccess(devices[p], 0);
            (void)hipGetLastError();
        };
        /* (also when ranks share a device -- the rehearsal on one GPU -- so that the tests run the path a node runs) */
        std::vector<std::thread> th;
        for (int r = 1; r < T; r++) th.emplace_back(setup, r);
        setup(0);
        for (std::thread &t : th) t.join();
        for (int r = 0; r < T && rc == SIFT3D_OK; r++)
            if (rcs[(size_t)r] != SIFT3D_OK) {
                rc = rcs[(size_t)r];
                snprintf(errbuf, sizeof errbuf, "%s", errs[(size_t)r].c_str());
            }
    }
    if (status_out) *status_out = rc;
    if (rc != SIFT3D_OK) {
        sift3d_zslab_destroy(h);
        return fail(errbuf);
    }
    {
        std::vector<int> devs((size_t)T);
        for (int r = 0; r < T; r++) devs[(size_t)r] = h->R[(size_t)r].dev;
        h->crew.start(T - 1, [devs](int r) { (void)hipSetDevice(devs[(size_t)r]); });
    }
    return h;
}

extern "C" sift3d_zslab *sift3d_zslab_create(const int *devices, int n_devices, int64_t nx, int64_t ny, int64_t nz, char *err, int64_t err_len)
{
    return zslab_create_impl(devices, n_devices, nx, ny, nz, err, err_len, nullptr);
}

/* vol: the whole volume in host memory, uploaded slab by slab inside the call -- or NULL: every rank's input slices are on
 * its device already (sift3d_zslab_set_volume).  out: the merged records malloc'ed for the caller -- or, with out == NULL,
 * *view: the handle's own merge buffer, valid until the handle's next call. */
static int zslab_extract_impl(sift3d_zslab *h, const float *vol, float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                              sift3d_feature **out, const sift3d_feature **view, int64_t *n_out, sift3d_zslab_stats *stats, char *err,
                              int64_t err_len)
{
    char errbuf[512] = "";
    int rc = SIFT3D_OK;
    int r = 0;
    if (err && err_len > 0) err[0] = 0;
    if (!h || (!out && !view) || !n_out || desc_mode < SIFT3D_DESC_SIFT || desc_mode > SIFT3D_DESC_NRRIEF) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "bad arguments");
        return SIFT3D_ERR_ARG;
    }
    if (!vol && !h->has_volume) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "no volume: call sift3d_zslab_set_volume first");
        return SIFT3D_ERR_ARG;
    }
    if (out) *out = nullptr;
    if (view) *view = nullptr;
    *n_out = 0;
    const zs_plan &plan = h->plan;
    const int64_t nx = plan.nx, ny = plan.ny, nz = plan.nz;
    std::vector<zs_rank> &R = h->R;
    const int S = h->n_slabs, T = (int)R.size();
    const int cr = h->coarse >= 0 ? h->coarse : 0; /* the rank that holds the octaves that are not sharded */
    const int K = plan.K;
    sift3d_zslab_stats st;
    memset(&st, 0, sizeof st);
    st.n_ranks = S;
    st.sharded_octaves = S > 1 ? K : 0;
    if (!h->tr) {
        std::vector<int> devs((size_t)S);
        for (int i = 0; i < S; i++) devs[(size_t)i] = R[(size_t)i].dev;
        char terr[400];
        h->tr = zs_transport_create(h->transport_want == SIFT3D_TRANSPORT_RCCL ? ZS_TRANSPORT_RCCL : ZS_TRANSPORT_PEER, h->transport_flags, devs.data(), S,
                                    terr, sizeof terr);
        if (!h->tr) {
            if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", terr);
            return SIFT3D_ERR_COMM;
        }
    }
    st.transport = zs_transport_kind(h->tr) == ZS_TRANSPORT_RCCL ? SIFT3D_TRANSPORT_RCCL : SIFT3D_TRANSPORT_PEER_COPY;
    st.transport_fell_back = zs_transport_fell_back(h->tr);
    st.rccl_version = zs_transport_version(h->tr);
    st.comm_sets = zs_transport_comm_sets(h->tr);
    st.resident_volume = vol ? 0 : 1;
    std::vector<int64_t> nrecs((size_t)T, 0);
    std::vector<std::vector<int>> shift; /* per rank: where its records of a group go in the merged list (alive until the streams are drained) */
    std::vector<int> rrc((size_t)T, SIFT3D_OK); /* what a rank's step reports (ZR_*), looked at when the step's ranks are back */
    std::vector<zs_errline> rerr((size_t)T);
    std::vector<zs_tally> tally((size_t)T);
    auto crew_failed = [&]() -> bool {
        for (int i = 0; i < T; i++)
            if (rrc[(size_t)i] != SIFT3D_OK) {
                rc = rrc[(size_t)i];
                snprintf(errbuf, sizeof errbuf, "%s", rerr[(size_t)i].b);
                return true;
            }
        return false;
    };
    const auto wall0 = std::chrono::steady_clock::now();

    /* sigma schedule, MultiScale.cpp:288-294,369,526-527 (float arithmetic as there) */
    float sigma_init = 0.5f;
    if (initial_image_scale > 0) sigma_init /= initial_image_scale;
    const float factor = (float)pow(2.0, 1.0 / (double)3);
    const float extra0 = sqrtf(1.6f * 1.6f - sigma_init * sigma_init);
    float extras[5], sig[6];
    {
        float sg = 1.6f;
        sig[0] = sg;
        for (int j = 0; j < 5; j++) {
            extras[j] = sg * sqrtf(factor * factor - 1.0f);
            sg *= factor;
            sig[j + 1] = sg;
        }
    }
    std::vector<float *> next0((size_t)T, nullptr); /* level 0 of the next octave per rank */
    float fscale = 1.0f;

    auto extraction = [&]() { /* returns early on a failure (rc / errbuf set); what follows it is done either way */
    /* ---- input slabs, level 0 of octave 0: every rank by its own host thread (an upload from pageable memory is a chain of
     * staged copies the calling thread takes part in: S of them side by side) ---- */
    h->crew.run(T, [&](int r) {
        zs_rank &q = R[(size_t)r];
        int64_t i0 = 0, i1 = nz;
        if (S > 1 && r < S) plan.input_range(r, i0, i1);
        ZR_HIP(hipSetDevice(q.dev));
        q.levels.assign(plan.oct.size() * 3, sift3d_level());
        ZR_RC(cand_reset(q.c));
        timing_begin(q.c);
        if (r >= S) return; /* the rank of the gathered octaves has no slab of the input */
        /* level 0 = initial blur of the input, on slab +- 8 from input slab +- 16 */
        const int64_t XY = nx * ny;
        float *din = vol ? q.alloc((i1 - i0) * XY) : q.vol_dev;
        if (!din) ZR_FAIL(SIFT3D_ERR_MEMORY, "rank %d: out of device memory", r);
        if (vol) ZR_HIP(hipMemcpyAsync(din, vol + i0 * XY, sizeof(float) * (size_t)((i1 - i0) * XY), hipMemcpyHostToDevice, q.c->stream));
        int64_t z0 = 0, z1 = nz;
        if (S > 1) plan.slab(r, 0, z0, z1);
        const bool lo = S > 1 && r > 0, hi = S > 1 && r < S - 1;
        const int64_t e0 = lo ? std::max<int64_t>(0, z0 - ZS_HALO) : z0, e1 = hi ? std::min<int64_t>(nz, z1 + ZS_HALO) : z1;
        const int64_t c0 = lo ? std::max(e0, z0 - ZS_BLUR) : e0, c1 = hi ? std::min(e1, z1 + ZS_BLUR) : e1;
        float *l0 = q.alloc((e1 - e0) * XY);
        if (!l0) ZR_FAIL(SIFT3D_ERR_MEMORY, "rank %d: out of device memory", r);
        /* Level 0 = the initial blur of the input, wanted on slab +- 8 (planes [c0, c1)) from input slab +- 16 (planes [i0, i1)).
         * Where the blur has its windowed form it writes exactly those planes straight into the level buffer: seen from the
         * input's plane numbering the buffer starts i0 - e0 planes before its own first plane (i0 >= e0: the input reaches 16
         * slices beyond the slab, the buffer 32), and only the window is written.  Otherwise: the whole input slab into a
         * scratch volume, then a copy of the planes that are exact (rounds 2 - 4). */
        if (i0 >= e0 && blur_window_supported(nx, ny, extra0, 0.01f)) {
            ZR_RC(blur_window_dev(q.c, din, l0 + (i0 - e0) * XY, nullptr, nx, ny, i1 - i0, c0 - i0, c1 - i0, extra0, 0.01f));
        } else {
            float *tmp = q.alloc((i1 - i0) * XY);
            if (!tmp) ZR_FAIL(SIFT3D_ERR_MEMORY, "rank %d: out of device memory", r);
            ZR_RC(blur_dev(q.c, din, tmp, nullptr, nx, ny, i1 - i0, extra0, 0.01f));
            ZR_HIP(hipMemcpyAsync(l0 + (c0 - e0) * XY, tmp + (c0 - i0) * XY, sizeof(float) * (size_t)((c1 - c0) * XY), hipMemcpyDeviceToDevice, q.c->stream));
        }
        next0[r] = l0;
    });
    if (crew_failed()) return;

    /* ---- one octave.  A sharded one is called from the calling thread and walks its steps with the crew; one that is not sharded
     * belongs to rank cr alone and is called from cr's own thread, inside the step that starts the per-keypoint stage (below): its
     * "steps" then run inline.  Failures go to the slots (rrc / rerr) either way. ---- */
    auto octave = [&](int o) {
        const int64_t X = plan.oct[(size_t)o][0], Y = plan.oct[(size_t)o][1], zo = plan.oct[(size_t)o][2], XY = X * Y;
        const bool sharded = S > 1 && o < K;
        const int r0 = sharded ? 0 : cr, r1 = sharded ? S : cr + 1; /* the ranks that hold a part of this octave */
        int r = r0;                                                  /* (ZR_* in this lambda's own statements: the slot of the rank at hand) */
        auto step = [&](auto &&f) { /* f(rank) for the octave's ranks */
            if (sharded) h->crew.run(S, f); else f(cr);
        };
        auto step_failed = [&]() -> bool { /* (an octave of cr's own runs beside other ranks' steps: it looks at its own slot only) */
            if (!sharded) return rrc[(size_t)cr] != SIFT3D_OK;
            for (int i = 0; i < T; i++)
                if (rrc[(size_t)i] != SIFT3D_OK) return true;
            return false;
        };
        /* As on one device (run_pipeline): D_0 is read as L_0 - L_1 around the extrema of D_1, and L_5 -- hence D_4 -- is
         * filtered only around the candidates of D_3, from L_4.  A slab then blurs four levels instead of five and exchanges
         * four halos per octave instead of five; the third extrema phase reads L_4 nine slices beyond a candidate, so L_4's
         * halo is refreshed nine slices deep instead of eight.  Rows that are not whole 16-byte vectors keep every level
         * stored (the extrema kernels of such rows take stored levels only), as does sift3d_zslab_set_tuning(SIFT3D_TUNE_LAZY_LEVELS, 0). */
        float taps5[SIFT3D_MAX_TAPS];
        const int ntaps5 = sift3d_gauss_taps(extras[4], 0.01f, taps5);
        const bool lazy = ntaps5 == 2 * SIFT3D_FAST_MAX_R + 1 && X % 4 == 0 && X >= 8 && XY < (1ll << 29) && Y >= 3 && zo >= 3 && h->lazy_levels;
        const int nlev = lazy ? 4 : 5;
        for (r = r0; r < r1; r++) { /* (bump allocations from the rank's arena: nothing to share out) */
            zs_rank &q = R[(size_t)r];
            if (sharded) plan.slab(r, o, q.z0, q.z1); else { q.z0 = 0; q.z1 = zo; }
            q.lo = sharded && r > 0;
            q.hi = sharded && r < S - 1;
            q.e0 = q.lo ? std::max<int64_t>(0, q.z0 - ZS_HALO) : q.z0;
            q.e1 = q.hi ? std::min<int64_t>(zo, q.z1 + ZS_HALO) : q.z1;
            ZR_HIP(hipSetDevice(q.dev));
            q.L[0] = next0[(size_t)r];
            for (int j = 1; j < 6; j++) q.L[j] = j <= nlev ? q.alloc((q.e1 - q.e0) * XY) : nullptr;
            for (int j = 0; j < 5; j++) q.D[j] = (lazy && (j == 0 || j == 4)) ? nullptr : q.alloc((q.e1 - q.e0) * XY);
            for (int j = 1; j <= nlev; j++)
                if (!q.L[j] || (!q.D[j - 1] && !(lazy && j == 1))) ZR_FAIL(SIFT3D_ERR_MEMORY, "rank %d: out of device memory", r);
        }
        r = r0;
        for (int j = 1; j <= nlev; j++) {
            const int64_t hb = (lazy && j == 4) ? ZS_BLUR + 1 : ZS_BLUR; /* slices of this level's halo refreshed from the neighbours */
            /* Boundary bands first (round 3).  What a neighbour fetches of this level are a rank's own first and last hb
             * slices.  Where the blur has a windowed form (the one-launch kernel: rows of whole 16-byte vectors, at most 17
             * taps) a rank filters those two bands first, records the event its neighbours' copies wait for, and then filters
             * its interior while the bands travel on the receivers' halo streams; its own halo slices are not computed at
             * all, they arrive.  The next level's launches wait for the arrivals, which by then have had the whole interior
             * launch to complete: no halo byte is waited for with an idle device unless the link is slower than the
             * interior.  Without the windowed form (other row lengths): the level on slab +- 8 in one piece, then the
             * exchange, as in round 2. */
            const bool banded = sharded && blur_window_supported(X, Y, extras[j - 1], 0.01f) && h->bands_first;
            /* -- the level's launches: every rank by its own thread (bands, event, interior).  The transfers of the bands are
             *    queued in the step after this one -- in host time behind the interior launch, on the device behind the event only -- */
            step([&](int r) {
                zs_rank &q = R[(size_t)r];
                const int64_t c0 = q.lo ? std::max(q.e0, q.z0 - ZS_BLUR) : q.e0, c1 = q.hi ? std::min(q.e1, q.z1 + ZS_BLUR) : q.e1;
                const int64_t a = c0 - q.e0, b = c1 - q.e0, nzl = q.e1 - q.e0;
                ZR_HIP(hipSetDevice(q.dev));
                if (banded && (q.lo || q.hi)) {
                    if (q.lo) ZR_RC(blur_window_dev(q.c, q.L[j - 1], q.L[j], q.D[j - 1], X, Y, nzl, q.z0 - q.e0, q.z0 - q.e0 + hb, extras[j - 1], 0.01f));
                    if (q.hi) ZR_RC(blur_window_dev(q.c, q.L[j - 1], q.L[j], q.D[j - 1], X, Y, nzl, q.z1 - q.e0 - hb, q.z1 - q.e0, extras[j - 1], 0.01f));
                    ZR_HIP(hipEventRecord(q.ev_level, q.c->stream)); /* the bands are final: the neighbours may fetch them */
                    const int64_t w0 = q.lo ? q.z0 - q.e0 + hb : 0, w1 = q.hi ? q.z1 - q.e0 - hb : q.e1 - q.e0;
                    ZR_RC(blur_window_dev(q.c, q.L[j - 1], q.L[j], q.D[j - 1], X, Y, nzl, w0, w1, extras[j - 1], 0.01f)); /* the interior, while the bands travel */
                    if (j == 3) ZR_HIP(hipEventRecord(q.ev_l3, q.c->stream));
                } else {
                    /* level j on slab +- 8 (clipped to the buffer: at a face of the whole volume the buffer ends at the face,
                     * which is what makes the zero border exact), D_{j-1} fused */
                    ZR_RC(blur_dev(q.c, q.L[j - 1] + a * XY, q.L[j] + a * XY, q.D[j - 1] ? q.D[j - 1] + a * XY : nullptr, X, Y, b - a, extras[j - 1], 0.01f));
                    ZR_HIP(hipEventRecord(q.ev_level, q.c->stream));
                    if (j == 3) ZR_HIP(hipEventRecord(q.ev_l3, q.c->stream));
                }
            });
            if (step_failed()) return;
            if (!sharded) continue; /* one rank holds the whole octave: no halo, nothing to redo */
            /* -- the hb-slice halo of the new level from the two neighbours (their own slices, exact), queued behind the sender's
             *    event -- on the receiver's halo stream beside its interior launch (bands first), or on its main stream; then the
             *    fused DoG redone on the halo slices; after level 3, what is left of the halos of L1..L3 (below).
             *    Peer copies are queued by the RECEIVER's thread (a copy into rank r is an operation of r's streams); RCCL's sends and
             *    receives of one step belong in one group of one thread, so the calling thread queues them for all ranks. -- */
            auto incoming = [&](int r) -> int { /* the transfers into rank r of this level; 0 or -1 (the transport holds the message) */
                zs_rank &q = R[(size_t)r];
                const size_t bytes = sizeof(float) * (size_t)(hb * XY);
                hipStream_t hs = banded ? q.halo_stream : q.c->stream;
                if (q.lo) { /* global slices [z0 - hb, z0): the lower neighbour's last own slices */
                    zs_rank &p = R[(size_t)r - 1];
                    if (zs_xfer(h->tr, 0, r - 1, p.L[j] + (q.z0 - hb - p.e0) * XY, banded ? p.halo_stream : p.c->stream, p.ev_level, r,
                                q.L[j] + (q.z0 - hb - q.e0) * XY, hs, (size_t)(hb * XY)) != 0) return -1;
                    tally[(size_t)r].critical += (int64_t)bytes;
                    if (banded) tally[(size_t)r].hidden += (int64_t)bytes;
                    tally[(size_t)r].exchanges++;
                }
                if (q.hi) { /* global slices [z1, z1 + hb): the upper neighbour's first own slices */
                    zs_rank &p = R[(size_t)r + 1];
                    if (zs_xfer(h->tr, 0, r + 1, p.L[j] + (q.z1 - p.e0) * XY, banded ? p.halo_stream : p.c->stream, p.ev_level, r,
                                q.L[j] + (q.z1 - q.e0) * XY, hs, (size_t)(hb * XY)) != 0) return -1;
                    tally[(size_t)r].critical += (int64_t)bytes;
                    if (banded) tally[(size_t)r].hidden += (int64_t)bytes;
                    tally[(size_t)r].exchanges++;
                }
                return 0;
            };
            /* L1..L3 are final after level 3: what is left of their halos, on the copy stream, while L4 and the extrema passes run.
             * Two steps (round 5; rounds 1 - 4: one batch of 3 x 24 slices, and the NEXT OCTAVE waited for all of it):
             *   step 0  the eight slices of L3 beyond +- 8 that the subsample reads (slab +- 16): all the next octave waits for;
             *   step 1  what only patches reach: L1 and L2 from 8, L3 from 16, as deep as a patch of that level can reach
             *           (ZS_PATCH_REACH: 19 / 23 / 28 slices, not 32) -- waited for before the per-keypoint stage, i.e. with the
             *           whole rest of the pyramid to arrive in.
             * Both steps travel on channel 1 in this order; a transport orders a channel's steps. */
            auto incoming_deferred = [&](int r, int step) -> int {
                zs_rank &q = R[(size_t)r];
                if (!q.lo && !q.hi) return 0;
                for (int l = step == 0 ? 3 : 1; l <= 3; l++) {
                    const int64_t from = (step == 0 || l < 3) ? ZS_BLUR : ZS_SUB, to = step == 0 ? ZS_SUB : ZS_PATCH_REACH[l];
                    if (q.lo) {
                        zs_rank &p = R[(size_t)r - 1];
                        const int64_t s0 = std::max(q.e0, q.z0 - to), s1 = q.z0 - from;
                        if (s1 > s0) {
                            if (zs_xfer(h->tr, 1, r - 1, p.L[l] + (s0 - p.e0) * XY, p.copy_stream, p.ev_l3, r, q.L[l] + (s0 - q.e0) * XY, q.copy_stream,
                                        (size_t)((s1 - s0) * XY)) != 0) return -1;
                            tally[(size_t)r].deferred += (int64_t)sizeof(float) * (s1 - s0) * XY;
                            if (step == 0) tally[(size_t)r].subsample += (int64_t)sizeof(float) * (s1 - s0) * XY;
                        }
                    }
                    if (q.hi) {
                        zs_rank &p = R[(size_t)r + 1];
                        const int64_t s0 = q.z1 + from, s1 = std::min(q.e1, q.z1 + to);
                        if (s1 > s0) {
                            if (zs_xfer(h->tr, 1, r + 1, p.L[l] + (s0 - p.e0) * XY, p.copy_stream, p.ev_l3, r, q.L[l] + (s0 - q.e0) * XY, q.copy_stream,
                                        (size_t)((s1 - s0) * XY)) != 0) return -1;
                            tally[(size_t)r].deferred += (int64_t)sizeof(float) * (s1 - s0) * XY;
                            if (step == 0) tally[(size_t)r].subsample += (int64_t)sizeof(float) * (s1 - s0) * XY;
                        }
                    }
                }
                tally[(size_t)r].exchanges++;
                return 0;
            };
            auto after_arrival = [&](int r) { /* rank r's side of the step once its transfers are queued */
                zs_rank &q = R[(size_t)r];
                ZR_HIP(hipSetDevice(q.dev));
                if (banded && (q.lo || q.hi)) { /* the main stream goes on behind the arrivals */
                    ZR_COMM(hipEventRecord(q.ev_halo, q.halo_stream));
                    ZR_COMM(hipStreamWaitEvent(q.c->stream, q.ev_halo, 0));
                }
                const int64_t a = (q.lo ? std::max(q.e0, q.z0 - ZS_BLUR) : q.e0) - q.e0, b = (q.hi ? std::min(q.e1, q.z1 + ZS_BLUR) : q.e1) - q.e0;
                if (q.D[j - 1] && q.lo && q.z0 - q.e0 > a)
                    ZR_HIP(sift3d_launch_dog(q.c->stream, q.L[j - 1] + a * XY, q.L[j] + a * XY, q.D[j - 1] + a * XY, (q.z0 - q.e0 - a) * XY));
                if (q.D[j - 1] && q.hi && b > q.z1 - q.e0)
                    ZR_HIP(sift3d_launch_dog(q.c->stream, q.L[j - 1] + (q.z1 - q.e0) * XY, q.L[j] + (q.z1 - q.e0) * XY, q.D[j - 1] + (q.z1 - q.e0) * XY, (b - (q.z1 - q.e0)) * XY));
            };
            auto after_deferred = [&](int r, int step) {
                zs_rank &q = R[(size_t)r];
                if (!q.lo && !q.hi) return;
                ZR_HIP(hipSetDevice(q.dev));
                if (step == h->patch_wait) ZR_HIP(hipEventRecord(q.ev_patch, q.copy_stream)); /* the subsample's slices are in */
                if (step == 1 && h->poison_halo) /* tests: a patch that reaches beyond what was fetched reads NaN and shows in the records */
                    for (int l = 1; l <= 3; l++) {
                        const int64_t reach = std::max(ZS_PATCH_REACH[l] - (h->poison_halo - 1), l == 3 ? ZS_SUB : ZS_BLUR);
                        if (q.lo && q.z0 - reach > q.e0)
                            ZR_HIP(hipMemsetD32Async((hipDeviceptr_t)q.L[l], 0x7FC00000, (size_t)((q.z0 - reach - q.e0) * XY), q.copy_stream));
                        if (q.hi && q.e1 > q.z1 + reach)
                            ZR_HIP(hipMemsetD32Async((hipDeviceptr_t)(q.L[l] + (q.z1 + reach - q.e0) * XY), 0x7FC00000, (size_t)((q.e1 - q.z1 - reach) * XY), q.copy_stream));
                    }
            };
            if (zs_transport_kind(h->tr) == ZS_TRANSPORT_RCCL) {
                r = 0; /* (the calling thread's own statements: rank 0's slot) */
                ZR_X(zs_xfer_begin(h->tr));
                for (int i = 0; i < S; i++) ZR_X(incoming(i));
                ZR_X(zs_xfer_end(h->tr)); /* RCCL queues the step's sends and receives here: what follows is behind them */
                h->crew.run(S, after_arrival);
                if (step_failed()) return;
                if (j == 3)
                    for (int part = 0; part < 2; part++) {
                        ZR_X(zs_xfer_begin(h->tr));
                        for (int i = 0; i < S; i++) ZR_X(incoming_deferred(i, part));
                        ZR_X(zs_xfer_end(h->tr));
                        for (int i = 0; i < S; i++) after_deferred(i, part);
                        if (step_failed()) return;
                    }
            } else {
                h->crew.run(S, [&](int r) {
                    ZR_X(incoming(r));
                    after_arrival(r);
                    if (rrc[(size_t)r] != SIFT3D_OK || j != 3) return;
                    for (int part = 0; part < 2; part++) {
                        ZR_X(incoming_deferred(r, part));
                        after_deferred(r, part);
                        if (rrc[(size_t)r] != SIFT3D_OK) return;
                    }
                });
                if (step_failed()) return;
            }
        }
        /* extrema of the rank's own slices; the level table in whole-volume terms.  Round 5: on the context's extrema stream,
         * behind everything the main stream holds so far (this octave's levels, their halos, the DoG slices redone on them), so
         * that the passes of octave o run beside the levels of octave o + 1 as they do on one device (run_pipeline); the main
         * stream waits for them once, before the counts are read.  Then level 0 of the next octave where that is the rank's own
         * business (the next octave sharded too: slab +- 16 of L3 -> next slab +- 8). */
        const bool more = o + 1 < (int)plan.oct.size();
        const int64_t Xn = more ? plan.oct[(size_t)o + 1][0] : 0, Yn = more ? plan.oct[(size_t)o + 1][1] : 0, zn = more ? plan.oct[(size_t)o + 1][2] : 0, XYn = Xn * Yn;
        step([&](int r) {
            zs_rank &q = R[(size_t)r];
            ZR_HIP(hipSetDevice(q.dev));
            ZR_HIP(hipEventRecord(q.c->ev_oct[0], q.c->stream));
            ZR_HIP(hipStreamWaitEvent(q.c->ex_stream, q.c->ev_oct[0], 0));
            q.c->cand_stream = q.c->ex_stream;
            for (int l = 0; l < 3; l++) {
                const int id = o * 3 + l;
                level_job jb = {q.D[l], q.D[l + 1], q.D[l + 2], X, Y, q.e1 - q.e0, (int)(q.z0 - q.e0), (int)(q.z1 - q.e0), id, 0};
                if (lazy && l == 0) { /* the level below D_1 is L_0 - L_1 */
                    jb.dp = q.L[0];
                    jb.prev_b = q.L[1];
                }
                if (lazy && l == 2) { /* the level above D_3 is L_4 - blur(L_4) */
                    jb.dn = nullptr;
                    jb.next_g = q.L[4];
                    jb.next_ntaps = ntaps5;
                    for (int t = 0; t < ntaps5; t++) jb.next_taps[t] = taps5[t];
                }
                const int rc_ = cand_append(q.c, jb, true);
                if (rc_ != SIFT3D_OK) {
                    q.c->cand_stream = nullptr;
                    ZR_FAIL(rc_, "rank %d: %s", r, sift3d_last_error(q.c));
                }
                sift3d_level &lv = q.levels[(size_t)id];
                lv.img = q.L[l + 1]; lv.dogc = q.D[l + 1];
                lv.X = (int)X; lv.Y = (int)Y; lv.Z = (int)zo; lv.XP = (int)X;
                lv.sigma_h = sig[l]; lv.sigma_c = sig[l + 1]; lv.sigma_l = sig[l + 2];
                lv.octave_factor = fscale;
                lv.Zl = (int)(q.e1 - q.e0); lv.z_off = (int)q.e0; lv.pad = 0;
            }
            q.c->cand_stream = nullptr;
            if (q.lo || q.hi) ZR_HIP(hipStreamWaitEvent(q.c->stream, q.ev_patch, 0)); /* before the subsample reads L3 beyond +- 8: the first deferred step only */
            if (more && sharded && o + 1 < K) {
                int64_t n0, n1;
                plan.slab(r, o + 1, n0, n1);
                const int64_t ne0 = q.lo ? std::max<int64_t>(0, n0 - ZS_HALO) : n0, ne1 = q.hi ? std::min<int64_t>(zn, n1 + ZS_HALO) : n1;
                const int64_t s0 = q.lo ? std::max(ne0, n0 - ZS_BLUR) : ne0, s1 = q.hi ? std::min(ne1, n1 + ZS_BLUR) : ne1;
                float *nx0 = q.alloc((ne1 - ne0) * XYn);
                if (!nx0) ZR_FAIL(SIFT3D_ERR_MEMORY, "rank %d: out of device memory", r);
                ZR_HIP(sift3d_launch_subsample(q.c->stream, q.L[3] + (2 * s0 - q.e0) * XY, X, X, Y, 2 * (s1 - s0), nx0 + (s0 - ne0) * XYn, Xn));
                next0[(size_t)r] = nx0;
            }
        });
        if (step_failed()) return;
        fscale *= 2.0f;
        if (!more) return;
        r = r0;
        /* ---- level 0 of the next octave where it is not a rank's own business ---- */
        if (sharded && o + 1 < K) {
            /* (done above, rank by rank) */
        } else if (sharded) { /* last sharded octave: every rank subsamples exactly its slab, rank 0 assembles the whole octave */
            /* (the calling thread queues all of it: a few launches and S - 1 transfers, once per extraction.)  The octave lands in
             * the buffers of rank cr -- the gathered octaves' own rank, on the first device -- and is ordered in ITS main stream:
             * the first device's part is a launch of rank 0's stream, the others arrive through the transport as transfers to
             * transport rank 0 (= that device). */
            zs_rank &root = R[(size_t)cr];
            r = cr;
            ZR_HIP(hipSetDevice(root.dev));
            float *full = root.alloc(zn * XYn);
            if (!full) ZR_FAIL(SIFT3D_ERR_MEMORY, "rank %d: out of device memory", r);
            int64_t at = 0;
            struct gather_part { int rank; const float *src; int64_t at, t; };
            std::vector<gather_part> parts;
            for (r = 0; r < S; r++) {
                zs_rank &q = R[(size_t)r];
                const int64_t t = std::min<int64_t>((q.z1 - q.z0) / 2, zn - at); /* an odd last slice of the whole volume is dropped, as in the serial code */
                if (t <= 0) continue;
                ZR_HIP(hipSetDevice(q.dev));
                float *part = r == 0 ? full + at * XYn : q.alloc(t * XYn);
                if (!part) ZR_FAIL(SIFT3D_ERR_MEMORY, "rank %d: out of device memory", r);
                ZR_HIP(sift3d_launch_subsample(q.c->stream, q.L[3] + (q.z0 - q.e0) * XY, X, X, Y, 2 * t, part, Xn));
                ZR_HIP(hipEventRecord(q.ev_level, q.c->stream));
                if (r > 0) {
                    parts.push_back({r, part, at, t});
                } else {
                    ZR_HIP(hipSetDevice(root.dev));
                    ZR_HIP(hipStreamWaitEvent(root.c->stream, q.ev_level, 0)); /* same device, another rank's stream */
                }
                at += t;
            }
            r = cr;
            ZR_X(zs_xfer_begin(h->tr)); /* every other device's part of the octave, ordered in the receiving rank's main stream */
            for (const gather_part &g : parts) {
                zs_rank &q = R[(size_t)g.rank];
                ZR_X(zs_xfer(h->tr, 0, g.rank, g.src, q.c->stream, q.ev_level, 0, full + g.at * XYn, root.c->stream, (size_t)(g.t * XYn)));
                st.gather_bytes += (int64_t)sizeof(float) * g.t * XYn;
            }
            ZR_X(zs_xfer_end(h->tr));
            if (at != zn) ZR_FAIL(SIFT3D_ERR_ARG, "slab plan does not tile octave %d (%lld of %lld slices)", o + 1, (long long)at, (long long)zn);
            next0[(size_t)cr] = full;
        } else { /* not sharded: rank cr alone */
            zs_rank &q = R[(size_t)cr];
            r = cr;
            ZR_HIP(hipSetDevice(q.dev));
            float *nx0 = q.alloc(zn * XYn);
            if (!nx0) ZR_FAIL(SIFT3D_ERR_MEMORY, "rank %d: out of device memory", r);
            ZR_HIP(sift3d_launch_subsample(q.c->stream, q.L[3], X, X, Y, zo, nx0, Xn));
            next0[(size_t)cr] = nx0;
        }
    };
    /* ---- the sharded octaves, step by step across the ranks (the others follow inside the per-keypoint stage's first step) ---- */
    const int n_sharded = S > 1 ? std::min<int>(K, (int)plan.oct.size()) : 0;
    for (int o = 0; o < n_sharded; o++) {
        octave(o);
        if (crew_failed()) return;
    }

    /* ---- per-keypoint stage: every rank by its own host thread, in ONE step of the crew.  A slab's rank requests and awaits its
     * extrema count, sorts, queues the keypoint kernel and reads back its records per (level, is_max) group; when every SLAB's rank
     * has those (a barrier among them inside the step) the places of their records in the merged list are known -- group by group,
     * rank by rank -- and every rank's descriptor kernel stores into its own.  The rank of the octaves that are not sharded takes
     * no part in that: its records come last in the list whatever the slabs' counts are, so it builds its octaves and runs its
     * per-keypoint stage into its own pinned buffer, and its few hundred records are appended at the end.  (First form of this
     * round: it was a rank like the others -- and every slab's descriptor launch waited for the chain of small launches that
     * builds the coarse octaves, which on a device shared with a keypoint kernel ends long after that kernel.) ---- */
    {
        std::vector<int64_t> ncands((size_t)T, 0);
        std::vector<double> queued_ms((size_t)T, 0.0); /* when a rank's thread had queued its pyramid, extrema passes and count request */
        std::vector<std::vector<int>> cnt((size_t)S, std::vector<int>(SIFT3D_GROUPS, 0));
        std::atomic<int> arrived{0};    /* slabs' ranks that have their counts -- or have failed */
        std::atomic<int> list_state{0}; /* 1: the merged list holds the slabs' records' places; -1: a rank failed, nobody stores */
        int64_t slab_total = 0;
        double layout_ms = 0.0;
        shift.assign((size_t)S, std::vector<int>(SIFT3D_GROUPS, 0)); /* (declared with the function's vectors: the uploads below read it) */
        auto spin_until = [](auto &&cond) {
            for (int spins = 0; !cond(); spins++)
                if (spins < 20000) __builtin_ia32_pause(); else std::this_thread::yield();
        };
        auto count_side = [&](int r, bool want_groups) { /* everything up to and including the keypoint kernel */
            zs_rank &q = R[(size_t)r];
            ZR_HIP(hipSetDevice(q.dev));
            /* the patch halos of every sharded octave (the second deferred step: the copy stream, in order) must be in before
             * the keypoint kernel samples them: everything the copy stream holds is behind this event */
            ZR_HIP(hipEventRecord(q.ev_l3, q.copy_stream)); /* (ev_l3 is free here: its last use was the last octave's sends) */
            ZR_HIP(hipStreamWaitEvent(q.c->stream, q.ev_l3, 0));
            ZR_HIP(hipEventRecord(q.c->ev_oct[1], q.c->ex_stream)); /* every extrema pass of the run is behind this */
            ZR_HIP(hipStreamWaitEvent(q.c->stream, q.c->ev_oct[1], 0));
            ZR_RC(cand_count_queue(q.c));
            queued_ms[(size_t)r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
            ZR_RC(cand_finalize(q.c, &ncands[(size_t)r]));
            describe_want_group_counts(q.c, want_groups);
            /* the places need the whole list's counts before the one descriptor launch: one chunk for this call only (the knob the
             * caller set on the handle is put back, as sift3d_describe_dev_counts does) */
            const int kp_chunks = q.c->tune[SIFT3D_TUNE_KP_CHUNKS];
            if (want_groups) q.c->tune[SIFT3D_TUNE_KP_CHUNKS] = 1;
            const int rc_desc = describe_queue(q.c, q.levels, ncands[(size_t)r], desc_mode, eig_thres, size_factor, false);
            q.c->tune[SIFT3D_TUNE_KP_CHUNKS] = kp_chunks;
            ZR_RC(rc_desc);
            if (!want_groups) return;
            /* this rank's records per group (level, is_max): the merged list is, group by group, a run of every rank in rank order --
             * within a group slabs are in z order, so rank order is the serial raster order */
            const int *hc = nullptr;
            int64_t tr = 0;
            ZR_RC(describe_group_counts(q.c, &hc, &tr));
            cnt[(size_t)r].assign(hc, hc + SIFT3D_GROUPS);
            nrecs[(size_t)r] = tr;
        };
        h->crew.run(T, [&](int r) {
            zs_rank &q = R[(size_t)r];
            if (h->coarse >= 0 && r == h->coarse) { /* the octaves that are not sharded: queued by this rank's own thread while the slabs' ranks go on */
                for (int o = n_sharded; o < (int)plan.oct.size(); o++) {
                    octave(o);
                    if (rrc[(size_t)r] != SIFT3D_OK) return;
                }
                count_side(r, false);
                if (rrc[(size_t)r] != SIFT3D_OK) return;
                ZR_RC(describe_launch(q.c)); /* into the context's own pinned buffers, as a context on its own does */
                ZR_RC(describe_finish(q.c, &nrecs[(size_t)r]));
                return;
            }
            if (S == 1) /* one rank, nothing sharded: the whole pyramid is this rank's */
                for (int o = 0; o < (int)plan.oct.size(); o++) {
                    octave(o);
                    if (rrc[(size_t)r] != SIFT3D_OK) break;
                }
            if (rrc[(size_t)r] == SIFT3D_OK) count_side(r, true);
            arrived.fetch_add(1); /* with counts or with a failure: nobody waits for a rank that has given up */
            if (r == 0) {
                spin_until([&] { return arrived.load() == S; });
                const auto t0 = std::chrono::steady_clock::now();
                bool ok = true;
                for (int i = 0; i < S; i++) ok = ok && rrc[(size_t)i] == SIFT3D_OK;
                if (ok) {
                    for (int i = 0; i < S; i++) slab_total += nrecs[(size_t)i];
                    const int64_t room = h->list_room >= 0 ? h->list_room : slab_total / 8 + 4096; /* for the coarse octaves' records, appended at the end */
                    if (slab_total + (h->list_room >= 0 ? 0 : slab_total / 64) > h->merged_cap || !h->merged) { /* nothing stores into the list yet */
                        if (h->merged) (void)hipHostFree(h->merged);
                        h->merged = nullptr;
                        h->merged_cap = slab_total + room;
                        if (hipHostMalloc((void **)&h->merged, sizeof(sift3d_feature) * (size_t)(h->merged_cap ? h->merged_cap : 1),
                                          hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
                            h->merged = nullptr;
                            h->merged_cap = 0;
                            rrc[0] = SIFT3D_ERR_MEMORY;
                            snprintf(rerr[0].b, sizeof rerr[0].b, "out of pinned host memory for %lld merged records", (long long)slab_total);
                            ok = false;
                        }
                    }
                }
                layout_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                list_state.store(ok ? 1 : -1);
            } else {
                spin_until([&] { return list_state.load() != 0; });
            }
            if (list_state.load() < 0 || rrc[(size_t)r] != SIFT3D_OK) return;
            { /* this rank's places: its records of group g go behind those of groups < g of every slab and of group g of the slabs before it */
                int64_t pos = 0, local = 0;
                for (int g = 0; g < SIFT3D_GROUPS; g++)
                    for (int i = 0; i < S; i++) {
                        if (i == r) {
                            shift[(size_t)r][(size_t)g] = (int)(pos - local);
                            local += cnt[(size_t)i][(size_t)g];
                        }
                        pos += cnt[(size_t)i][(size_t)g];
                    }
            }
            ZR_HIP(hipSetDevice(q.dev));
            if (nrecs[(size_t)r] > 0) {
                sift3d_feature *dview = nullptr; /* the list as this rank's device sees it */
                ZR_HIP(hipHostGetDevicePointer((void **)&dview, h->merged, 0));
                ZR_RC(describe_placement(q.c, dview, shift[(size_t)r].data()));
            }
            ZR_RC(describe_launch(q.c));
            int64_t nrec = 0;
            ZR_RC(describe_finish(q.c, &nrec));
            if (nrec != nrecs[(size_t)r])
                ZR_FAIL(SIFT3D_ERR_DEVICE, "rank %d: %lld records where its groups add up to %lld", r, (long long)nrec, (long long)nrecs[(size_t)r]);
        });
        if (crew_failed()) return;
        const auto merge0 = std::chrono::steady_clock::now();
        const int64_t ncoarse = h->coarse >= 0 ? nrecs[(size_t)h->coarse] : 0;
        const int64_t total = slab_total + ncoarse;
        for (r = 0; r < T; r++) {
            st.n_extrema += ncands[(size_t)r];
            st.enqueue_ms = std::max(st.enqueue_ms, queued_ms[(size_t)r]);
            st.n_keypoints += R[(size_t)r].c->last.n_keypoints;
        }
        if (total > h->merged_cap) { /* the coarse octaves' records do not fit behind the slabs': a larger list, the slabs' part carried over (rare) */
            sift3d_feature *big = nullptr;
            const int64_t cap = total + total / 8 + 4096;
            if (hipHostMalloc((void **)&big, sizeof(sift3d_feature) * (size_t)cap, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
                rc = SIFT3D_ERR_MEMORY;
                snprintf(errbuf, sizeof errbuf, "out of pinned host memory for %lld merged records", (long long)total);
                return;
            }
            if (slab_total) memcpy(big, h->merged, sizeof(sift3d_feature) * (size_t)slab_total);
            if (h->merged) (void)hipHostFree(h->merged);
            h->merged = big;
            h->merged_cap = cap;
            st.list_grown = 1;
        }
        if (ncoarse) memcpy(h->merged + slab_total, R[(size_t)h->coarse].c->h_recs, sizeof(sift3d_feature) * (size_t)ncoarse);
        st.merge_ms = layout_ms + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - merge0).count();
        /* the list is complete where every rank's kernel put it */
        if (out) {
            sift3d_feature *res = (sift3d_feature *)malloc(sizeof(sift3d_feature) * (size_t)(total ? total : 1));
            if (!res) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "out of host memory"); return; }
            if (total) memcpy(res, h->merged, sizeof(sift3d_feature) * (size_t)total);
            *out = res;
        } else {
            *view = h->merged;
        }
        *n_out = total;
        st.n_records = total;
    }

    };
    extraction();
    if (rc != SIFT3D_OK) zs_xfer_abort(h->tr); /* a step that failed between its begin and its end leaves no group open */
    for (size_t i = 0; i < R.size(); i++) { /* everything queued has to be done before the buffers go back */
        zs_rank &q = R[i];
        hipSetDevice(q.dev);
        hipStreamSynchronize(q.c->stream);
        hipStreamSynchronize(q.c->ex_stream);
        q.c->cand_stream = nullptr;
        hipStreamSynchronize(q.copy_stream);
        hipStreamSynchronize(q.halo_stream);
    }
    for (size_t i = 0; i < R.size(); i++) {
        hipSetDevice(R[i].dev);
        R[i].recycle();
    }
    if (rc == SIFT3D_ERR_COMM) { /* the next extraction starts with fresh communicators (the streams are drained: nothing is in flight) */
        zs_transport_destroy(h->tr);
        h->tr = nullptr;
    }
    for (const zs_tally &t : tally) {
        st.halo_bytes_critical += t.critical;
        st.halo_bytes_hidden += t.hidden;
        st.halo_bytes_deferred += t.deferred;
        st.halo_bytes_subsample += t.subsample;
        st.exchanges += t.exchanges;
    }
    st.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (stats) *stats = st;
    if (rc != SIFT3D_OK && err && err_len > 0) snprintf(err, (size_t)err_len, "%s", errbuf);
    return rc;
}

extern "C" int sift3d_zslab_extract(sift3d_zslab *h, const float *vol, float initial_image_scale, int desc_mode, float eig_thres,
                                    float size_factor, sift3d_feature **out, int64_t *n_out, sift3d_zslab_stats *stats, char *err,
                                    int64_t err_len)
{
    if (!vol || !out) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "bad arguments");
        return SIFT3D_ERR_ARG;
    }
    return zslab_extract_impl(h, vol, initial_image_scale, desc_mode, eig_thres, size_factor, out, nullptr, n_out, stats, err, err_len);
}

/* The whole volume (host memory) cut into the ranks' input slices and uploaded, once; the extractions that follow
 * (sift3d_zslab_extract_resident) start from HBM, as sift3d_extract does after sift3d_set_volume. */
extern "C" int sift3d_zslab_set_volume(sift3d_zslab *h, const float *vol, char *err, int64_t err_len)
{
    if (err && err_len > 0) err[0] = 0;
    if (!h || !vol) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "bad arguments");
        return SIFT3D_ERR_ARG;
    }
    const zs_plan &plan = h->plan;
    const int S = h->n_slabs; /* (the rank of the gathered octaves has no slab of the input) */
    const int64_t XY = plan.nx * plan.ny;
    h->has_volume = false;
    /* every rank's slices by its own host thread: an upload from pageable memory is staged by the thread that asked for it */
    std::vector<int> rcs((size_t)S, SIFT3D_OK);
    std::vector<zs_errline> errs((size_t)S);
    h->crew.run(S, [&](int r) {
        zs_rank &q = h->R[(size_t)r];
        int64_t i0 = 0, i1 = plan.nz;
        if (S > 1) plan.input_range(r, i0, i1);
        hipError_t e = hipSetDevice(q.dev);
        if (e == hipSuccess && !q.vol_dev) e = hipMalloc((void **)&q.vol_dev, sizeof(float) * (size_t)((i1 - i0) * XY));
        if (e == hipSuccess) e = hipMemcpyAsync(q.vol_dev, vol + i0 * XY, sizeof(float) * (size_t)((i1 - i0) * XY), hipMemcpyHostToDevice, q.c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(q.c->stream); /* the caller's buffer is free again when this returns */
        if (e != hipSuccess) {
            snprintf(errs[(size_t)r].b, sizeof errs[0].b, "rank %d: input slices [%lld, %lld): %s", r, (long long)i0, (long long)i1, hipGetErrorString(e));
            rcs[(size_t)r] = e == hipErrorOutOfMemory ? SIFT3D_ERR_MEMORY : SIFT3D_ERR_DEVICE;
        }
    });
    for (int r = 0; r < S; r++)
        if (rcs[(size_t)r] != SIFT3D_OK) {
            if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", errs[(size_t)r].b);
            return rcs[(size_t)r];
        }
    h->has_volume = true;
    return SIFT3D_OK;
}

extern "C" int sift3d_zslab_extract_resident(sift3d_zslab *h, float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                                             const sift3d_feature **view, int64_t *n_out, sift3d_zslab_stats *stats, char *err, int64_t err_len)
{
    if (!view) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "bad arguments");
        return SIFT3D_ERR_ARG;
    }
    return zslab_extract_impl(h, nullptr, initial_image_scale, desc_mode, eig_thres, size_factor, nullptr, view, n_out, stats, err, err_len);
}

extern "C" int sift3d_extract_zslab_over(int transport, const int *devices, int n_devices, const float *vol, int64_t nx, int64_t ny, int64_t nz,
                                         float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                                         sift3d_feature **out, int64_t *n_out, sift3d_zslab_stats *stats, char *err, int64_t err_len)
{
    if (out) *out = nullptr;
    if (n_out) *n_out = 0;
    if (!vol || !out || !n_out || desc_mode < SIFT3D_DESC_SIFT || desc_mode > SIFT3D_DESC_NRRIEF ||
        (transport != SIFT3D_TRANSPORT_PEER_COPY && transport != SIFT3D_TRANSPORT_RCCL)) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "bad arguments");
        return SIFT3D_ERR_ARG;
    }
    int status = SIFT3D_ERR_ARG;
    sift3d_zslab *h = zslab_create_impl(devices, n_devices, nx, ny, nz, err, err_len, &status);
    if (!h) return status != SIFT3D_OK ? status : SIFT3D_ERR_MEMORY;
    h->transport_want = transport;
    const int rc = sift3d_zslab_extract(h, vol, initial_image_scale, desc_mode, eig_thres, size_factor, out, n_out, stats, err, err_len);
    sift3d_zslab_destroy(h);
    return rc;
}

extern "C" int sift3d_extract_zslab(const int *devices, int n_devices, const float *vol, int64_t nx, int64_t ny, int64_t nz,
                                    float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                                    sift3d_feature **out, int64_t *n_out, sift3d_zslab_stats *stats, char *err, int64_t err_len)
{
    return sift3d_extract_zslab_over(SIFT3D_TRANSPORT_PEER_COPY, devices, n_devices, vol, nx, ny, nz, initial_image_scale, desc_mode, eig_thres,
                                     size_factor, out, n_out, stats, err, err_len);
}

extern "C" void sift3d_zslab_set_transport_library(const char *path) { zs_transport_set_library(path); }
