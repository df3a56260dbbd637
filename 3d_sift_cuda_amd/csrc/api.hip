/*
 * api.hip -- the C-ABI of include/sift3d.h: context (device-resident pyramid
 * buffers, stream, timing events), operator-level entry points and the
 * scale-space / extraction pipeline that strings the kernels together.
 *
 * Schedule = msGeneratePyramidDOG3D_efficient (R/src_common/MultiScale.cpp:236-570,
 * R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/): initial
 * blur to sigma 1.6, then per octave the levels L1..L5 (sigma ratio 2^(1/3)),
 * DoG k = L_k - L_{k+1} for k = 0..4, extrema in DoG 1..3, keypoints sampled
 * from L_k, next octave seeded by the 2x2x2 mean of L_3.  The reference
 * recycles five buffers and validates "on the fly"; here the levels something
 * reads in full -- L0..L4 and D1..D3 of every octave -- stay resident in HBM
 * (13.4 N floats with the intermediates: 288 GB holds a 1024^3 volume five
 * times over), each produced by one fused x + y + z + DoG launch where the
 * volume fills the chip (three launches on coarse octaves) and read by one
 * extrema pass; D0, D4 and L5 are evaluated only around the candidates
 * (DESIGN.md section 4).  Nothing leaves the device between the upload of the
 * volume and the records, which the descriptor kernel stores straight into
 * pinned host memory.
 *
 * Layout of this file: context and tuning; timing; operator-level entry points;
 * candidate lists (reset / append / finalize); the per-keypoint stage in three
 * phases (describe_queue / _launch / _finish); run_pipeline; the slab building
 * blocks of the C-ABI; the one-process Z-slab driver (sift3d_zslab_*).
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sift3d_internal.h"
#include "zslab_transport.h"

#define SIFT3D_KP_MAX_CHUNKS 16
/* One chunk by default: measured at 512^3 (tools/kp_chunks.py, profiles/r03_kp_chunks.txt) 1: 10.33, 2: 10.31, 3: 10.47, 4: 10.47,
 * 6: 10.73, 8: 10.99, 16: 11.20 ms per extraction -- seven keypoint workgroups fill a CU's LDS (7 x 23 KB), so a descriptor
 * workgroup only becomes resident where a keypoint workgroup has retired, and the two kernels take turns instead of sharing. */
#define SIFT3D_KP_DEFAULT_CHUNKS 1
#define SIFT3D_D4TINY_FLOATS 32768 /* room for the octaves of at most SIFT3D_TINY_VOX voxels of one volume, pitched rows included */

struct timed_launch {
    int stage;
    hipEvent_t e0, e1;
    int ntaps;
    int64_t nvox;
    double bytes;
    float ms;
    float start_ms;
};

/* One detection level: which buffers, which dims, which slices to keep */
struct level_job {
    const float *dp, *dc, *dn;
    int64_t X, Y, Z; /* X is the row pitch of the buffers */
    int z_lo, z_hi;
    int lvl_id;
    int64_t Xl;      /* logical row length (0: same as X) */
    /* neighbour levels that are not stored (sift3d_extrema_lazy): the level below is dp - prev_b; the level above is
     * next_g - blur(next_g, next_taps), evaluated around the candidates only (dn is NULL then) */
    const float *prev_b = nullptr, *next_g = nullptr;
    float next_taps[2 * SIFT3D_FAST_MAX_R + 1] = {};
    int next_ntaps = 0;
};

struct octave_dims {
    int64_t X, Y, Z, off; /* dims and float offset of this octave inside every level buffer */
    int64_t XP;           /* row pitch: X rounded up to whole 16-byte vectors (the pad columns stay zero) */
};

struct sift3d_ctx {
    int device;
    hipStream_t stream;
    hipStream_t ex_stream;     /* extrema detection of an octave, overlapped with the blurs of the coarser octaves */
    hipStream_t cand_stream;   /* where cand_append launches: stream, or ex_stream inside run_pipeline */
    hipStream_t ex_stream2;    /* extrema of the octaves after the first */
    hipEvent_t ev_ex2[2];      /* levels of such an octave complete / its extrema launches complete */
    hipEvent_t ev_reset;       /* the counters of the extrema passes have been cleared (on ex_stream) */
    sift3d_survivor *surv2;    /* own-level list of that stream (the passes of one stream share a list, one after the other) */
    int64_t surv2_cap;
    int surv_sel;              /* which list cand_append uses: 0 = surv, 1 = surv2 */
    hipStream_t kp_stream;     /* descriptor launches of the chunked per-keypoint stage, beside the keypoint kernel of the next chunk */
    hipEvent_t ev_kpc[SIFT3D_KP_MAX_CHUNKS]; /* chunk i's keypoint kernel, scan and record map are complete */
    hipEvent_t ev_desc;        /* the descriptor launches on kp_stream are complete */
    unsigned long long *h_cnt0; /* pinned, 8 + SIFT3D_KP_MAX_CHUNKS words: [4..7] the small read-backs the host waits for (extrema counts,
                                 * keypoint count), [8..] the record totals of the chunks: a copy into pageable memory goes through a
                                 * staging buffer and costs tens of microseconds more */
    hipEvent_t ev_oct[2];      /* octave's DoG levels complete / extrema launches complete */
    hipEvent_t ev_fence[2];    /* ordering of the *_dev entry points with the legacy default stream (fence_in / fence_out) */
    bool own_stream;
    int64_t capN;   /* voxels of the largest volume */
    int64_t capTot; /* floats per level buffer: all octaves of a capN volume back to back */
    float *vol;   /* input volume */
    float *L[6];  /* Gaussian levels, every octave resident (octave o at offset off_o); L[5] is never stored and stays NULL */
    float *D[5];  /* DoG levels, same layout; D[4] is only allocated when an octave has to store its last DoG level in full
                   * (ensure_level_buffer): by default that level is evaluated around the candidates only */
    float *D4tiny; /* the last DoG level of the octaves that one workgroup builds whole (at most SIFT3D_TINY_VOX voxels each) */
    float *T[2];  /* x- and y-pass intermediates */
    float *d_taps;
    /* extrema as (key, value) pairs, unsorted (a) and sorted (b) */
    unsigned long long *keys_a, *keys_b;
    sift3d_cval *vals_a, *vals_b;
    int64_t cand_cap;
    unsigned long long *d_count; /* [0] validated extrema, [1] own-level survivors of the level in flight, [2] survivor overflow high-water mark */
    sift3d_survivor *surv;
    sift3d_survivor2 *list2[2];      /* extrema that passed the level below, waiting for the lazily evaluated level above: one list
                                      * per extrema stream (surv_sel) */
    int64_t list2_cap[2];
    unsigned long long *list2_counts; /* one length word per extrema pass (SIFT3D_SURV_SETS), zeroed with surv_counts */
    unsigned long long *surv_counts; /* segment counters of the own-level list: SIFT3D_SURV_SETS sets */
    int surv_set;                    /* next unused set since the last reset */
    int64_t surv_cap;
    int surv_div; /* own-level extrema expected per level: voxels / surv_div (+ slack); 1 after an overflow */
    void *sort_tmp;
    size_t sort_tmp_bytes;
    void *scan_tmp;
    size_t scan_tmp_bytes;
    sift3d_level *d_levels;
    sift3d_dkp *kps;
    float *patch0; /* identity-frame patches of the extrema, kps_cap x 1331 floats */
    int *sampler_tokens; /* per-CU counters of the descriptor kernel's sampling phase (zero whenever no kernel runs) */
    int *d_rec_base;     /* chunked per-keypoint stage: first record of chunk i (SIFT3D_KP_MAX_CHUNKS + 1 ints; [n] = total) */
    int *nrec, *offs; /* per-candidate record count and exclusive prefix */
    int64_t kps_cap;
    int *rec_kp, *rec_frame;
    int64_t recs_cap;       /* record slots of rec_kp / rec_frame: kps_cap * (1 + SIFT3D_MAX_FRAMES), the worst case */
    int64_t capT;           /* floats each of T[0], T[1] holds */
    int64_t hrecs_cap;      /* records the two pinned host buffers below hold: a few per candidate, grown when a run needs more */
    sift3d_feature *h_recs; /* pinned host memory the descriptor kernel stores its records into; reused from call to call */
    int *h_group;           /* per record: level id * 2 + is_max (pinned host) */
    sift3d_feature *d_hrecs; /* the device's addresses of the two */
    int *d_hgroup;
    struct {                /* the per-keypoint stage in flight (describe_queue / _launch / _finish) */
        sift3d_kp_params p;
        float taps5[SIFT3D_MAX_TAPS];
        int64_t ncand, nrec;
        int nchunks, launched;
        int64_t first[SIFT3D_KP_MAX_CHUNKS + 1];
    } kp;
    int dev_stop;           /* -DSIFT3D_DEV builds: sift3d_dev_set_stop */
    bool count_queued;      /* cand_count_queue ran and nothing was appended since */
    std::vector<struct level_job> jobs; /* extrema launches since the last reset (replayed if the buffer must grow) */
    int64_t nx, ny, nz;
    int64_t pad_nx, pad_ny, pad_nz; /* geometry the pad columns of the level buffers were last cleared for */
    bool has_volume;
    int max_octaves; /* 0: the reference's only stop rule (a dimension <= 2); n > 0: at most n octaves */
    int tune[SIFT3D_TUNE_COUNT]; /* sift3d_set_tuning */
    int64_t host_grows;          /* times describe_launch had to grow the pinned record buffers (tests) */
    bool lean;       /* a slab context: the caller owns the level buffers, none are allocated here */
    int timing; /* 0 off; 1 every launch bracketed by events; 2 only the blur launches of the finest octave */
    std::vector<timed_launch> launches;
    std::vector<hipEvent_t> pool;
    size_t pool_used;
    size_t resolved; /* launches whose events have been read */
    sift3d_timings last;
    char err[512];
};

static int set_err(sift3d_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(c, call)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return set_err((c), SIFT3D_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                                          \
    } while (0)

/* entry points that work in the context's own level buffers: not on a slab context, which has none */
#define NEED_LEVELS(c)                                                                                                  \
    do {                                                                                                                \
        if ((c) && (c)->lean) return set_err((c), SIFT3D_ERR_ARG, "%s needs a full context (sift3d_create), not a slab context", __func__); \
    } while (0)

extern "C" int sift3d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void sift3d_free(void *p) { free(p); }

extern "C" const char *sift3d_last_error(const sift3d_ctx *ctx) { return ctx ? ctx->err : "no context"; }

static void free_dev(sift3d_ctx *c)
{
    hipFree(c->vol);
    for (int i = 0; i < 6; i++) hipFree(c->L[i]);
    for (int i = 0; i < 5; i++) hipFree(c->D[i]);
    hipFree(c->D4tiny);
    hipFree(c->T[0]);
    hipFree(c->T[1]);
    hipFree(c->d_taps);
    hipFree(c->keys_a);
    hipFree(c->keys_b);
    hipFree(c->vals_a);
    hipFree(c->vals_b);
    hipFree(c->d_count);
    hipFree(c->surv);
    hipFree(c->surv2);
    hipFree(c->list2[0]);
    hipFree(c->list2[1]);
    hipFree(c->list2_counts);
    hipFree(c->surv_counts);
    hipFree(c->sort_tmp);
    hipFree(c->scan_tmp);
    hipFree(c->d_levels);
    hipFree(c->kps);
    hipFree(c->patch0);
    hipFree(c->sampler_tokens);
    hipFree(c->d_rec_base);
    hipFree(c->nrec);
    hipFree(c->offs);
    hipFree(c->rec_kp);
    hipFree(c->rec_frame);
    if (c->h_recs) hipHostFree(c->h_recs);
    if (c->h_group) hipHostFree(c->h_group);
}

/* octave list of a volume: halve while every dimension stays above 2 (MultiScale.cpp:359-360,546-556) */
static inline int64_t pitch_of(int64_t X) { return (X + 3) / 4 * 4; }

/* Inside the pipeline every octave is stored with rows padded to whole 16-byte vectors; the pad columns hold zeros
 * (what the blur reads outside the volume), so the vector kernels serve any row length. */
static std::vector<octave_dims> octave_list(int64_t X, int64_t Y, int64_t Z)
{
    std::vector<octave_dims> v;
    int64_t off = 0;
    while (X > 2 && Y > 2 && Z > 2 && v.size() < 32) {
        const int64_t XP = pitch_of(X);
        v.push_back({X, Y, Z, off, XP});
        off += ((XP * Y * Z + 63) / 64) * 64; /* keep every octave 256-byte aligned */
        X /= 2; Y /= 2; Z /= 2;
    }
    return v;
}

static int alloc_cands(sift3d_ctx *c, int64_t cap)
{
    hipFree(c->keys_a); hipFree(c->keys_b); hipFree(c->vals_a); hipFree(c->vals_b); hipFree(c->sort_tmp);
    c->keys_a = c->keys_b = nullptr;
    c->vals_a = c->vals_b = nullptr;
    c->sort_tmp = nullptr;
    c->cand_cap = cap;
    c->sort_tmp_bytes = sift3d_sort_temp_bytes(cap) + 256;
    bool ok = hipMalloc((void **)&c->keys_a, sizeof(unsigned long long) * (size_t)cap) == hipSuccess &&
              hipMalloc((void **)&c->keys_b, sizeof(unsigned long long) * (size_t)cap) == hipSuccess &&
              hipMalloc((void **)&c->vals_a, sizeof(sift3d_cval) * (size_t)cap) == hipSuccess &&
              hipMalloc((void **)&c->vals_b, sizeof(sift3d_cval) * (size_t)cap) == hipSuccess &&
              hipMalloc(&c->sort_tmp, c->sort_tmp_bytes) == hipSuccess;
    return ok ? SIFT3D_OK : SIFT3D_ERR_MEMORY;
}

static void destroy_sync_objects(sift3d_ctx *c)
{
    hipStream_t streams[] = {c->ex_stream, c->ex_stream2, c->kp_stream};
    for (hipStream_t st : streams)
        if (st) {
            hipStreamSynchronize(st);
            hipStreamDestroy(st);
        }
    hipEvent_t events[] = {c->ev_ex2[0], c->ev_ex2[1], c->ev_reset, c->ev_desc, c->ev_oct[0], c->ev_oct[1], c->ev_fence[0], c->ev_fence[1]};
    for (hipEvent_t e : events)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_kpc)
        if (e) hipEventDestroy(e);
    if (c->h_cnt0) hipHostFree(c->h_cnt0);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
}

/* lean: a slab context (sift3d_create_slab) -- the caller owns the level buffers; only the pass intermediates, the
 * candidate lists and the per-keypoint buffers live here */
static sift3d_ctx *ctx_create(int device, int64_t nx, int64_t ny, int64_t nz, bool lean)
{
    if (nx <= 0 || ny <= 0 || nz <= 0) return nullptr;
    int n = sift3d_device_count();
    if (device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    sift3d_ctx *c = new sift3d_ctx(); /* value-initialised: every pointer null, every count zero */
    c->device = device;
    c->own_stream = true;
    c->lean = lean;
    c->capN = pitch_of(nx) * ny * nz; /* floats of the largest volume, rows padded to whole vectors */
    c->surv_div = 64;
    c->tune[SIFT3D_TUNE_BLUR_FUSED] = 1;
    c->tune[SIFT3D_TUNE_LAZY_LEVELS] = 1;
    c->tune[SIFT3D_TUNE_TINY_OCTAVE] = 1;
    c->tune[SIFT3D_TUNE_SAMPLER_CAP] = 4;
    c->tune[SIFT3D_TUNE_BANDS_FIRST] = 1;
    c->tune[SIFT3D_TUNE_HOST_RECORDS] = 5;
    /* every octave of a capN volume, back to back: capN * (1 + 1/8 + 1/64 + ...) plus alignment */
    c->capTot = c->capN + c->capN / 7 + 4 * ny * nz + 64 * 34; /* + up to three pad columns per row of every coarser octave */
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    hipStream_t *streams[] = {&c->ex_stream, &c->ex_stream2, &c->kp_stream};
    for (hipStream_t *st : streams) ok = ok && hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t *events[] = {&c->ev_ex2[0], &c->ev_ex2[1], &c->ev_reset, &c->ev_desc, &c->ev_oct[0], &c->ev_oct[1], &c->ev_fence[0], &c->ev_fence[1]};
    for (hipEvent_t *e : events) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    for (hipEvent_t &e : c->ev_kpc) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_cnt0, sizeof(unsigned long long) * (8 + SIFT3D_KP_MAX_CHUNKS), hipHostMallocDefault) == hipSuccess;
    const size_t vb = sizeof(float) * (size_t)c->capN;
    const size_t tb = sizeof(float) * (size_t)c->capTot;
    /* nothing may depend on what hipMalloc hands back: the pad columns of pitched octaves are read as zeros.  The clears
     * go on the context's own stream and are waited for here: hipMemset runs on the null stream, which the context's
     * non-blocking streams are NOT ordered with, so it could still be wiping a buffer the first extraction already uses */
    if (!lean) {
        ok = ok && hipMalloc((void **)&c->vol, vb) == hipSuccess && hipMemsetAsync(c->vol, 0, vb, c->stream) == hipSuccess;
        for (int i = 0; i < 5 && ok; i++) ok = hipMalloc((void **)&c->L[i], tb) == hipSuccess && hipMemsetAsync(c->L[i], 0, tb, c->stream) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipMalloc((void **)&c->D[i], tb) == hipSuccess && hipMemsetAsync(c->D[i], 0, tb, c->stream) == hipSuccess;
        ok = ok && hipMalloc((void **)&c->D4tiny, sizeof(float) * SIFT3D_D4TINY_FLOATS) == hipSuccess &&
             hipMemsetAsync(c->D4tiny, 0, sizeof(float) * SIFT3D_D4TINY_FLOATS, c->stream) == hipSuccess;
    }
    /* the two pass intermediates of the three-launch blur: allocated here for the volumes the numbers are quoted on; a
     * context beyond 2^31 voxels allocates them when a blur first takes that form, sized for it (its full-size levels go
     * through the one-launch kernel, the coarse octaves need an eighth) -- 2 x 17 GB less at config C5's 2^32 voxels */
    if (c->capN <= SIFT3D_EAGER_T_FLOATS)
        for (int i = 0; i < 2 && ok; i++) {
            ok = hipMalloc((void **)&c->T[i], vb) == hipSuccess;
            if (ok) c->capT = c->capN;
        }
    ok = ok && hipMalloc((void **)&c->d_taps, sizeof(float) * SIFT3D_MAX_TAPS) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_count, sizeof(unsigned long long) * 4) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_levels, sizeof(sift3d_level) * 96) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->sampler_tokens, sizeof(int) * SIFT3D_CU_SLOTS) == hipSuccess &&
         hipMemsetAsync(c->sampler_tokens, 0, sizeof(int) * SIFT3D_CU_SLOTS, c->stream) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_rec_base, sizeof(int) * (SIFT3D_KP_MAX_CHUNKS + 1)) == hipSuccess;
    ok = ok && alloc_cands(c, c->capN / 32 + 8192) == SIFT3D_OK;
    c->surv_cap = c->capN / 8 + 65536; /* own-level extrema are ~0.3 % of the voxels on blob fields, ~1 % on noise */
    ok = ok && hipMalloc((void **)&c->surv, sizeof(sift3d_survivor) * (size_t)c->surv_cap) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->surv_counts, sizeof(unsigned long long) * SIFT3D_SURV_COUNTERS * SIFT3D_SURV_SETS) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->list2_counts, sizeof(unsigned long long) * SIFT3D_LIST2_COUNTERS * SIFT3D_SURV_SETS) == hipSuccess;
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess; /* the clears above are done before the context is handed out */
    if (!ok) {
        free_dev(c);
        destroy_sync_objects(c);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" sift3d_ctx *sift3d_create(int device, int64_t nx, int64_t ny, int64_t nz) { return ctx_create(device, nx, ny, nz, false); }

extern "C" sift3d_ctx *sift3d_create_slab(int device, int64_t nx, int64_t ny, int64_t nz_local) { return ctx_create(device, nx, ny, nz_local, true); }

extern "C" void sift3d_destroy(sift3d_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    free_dev(c);
    for (hipEvent_t e : c->pool) hipEventDestroy(e);
    destroy_sync_objects(c);
    delete c;
}

extern "C" int sift3d_set_tuning(sift3d_ctx *c, int knob, int value)
{
    if (!c) return SIFT3D_ERR_ARG;
    static const int lo[SIFT3D_TUNE_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 0},
                     hi[SIFT3D_TUNE_COUNT] = {2, 4096, 2, 1, 1, 64, SIFT3D_KP_MAX_CHUNKS, 1, 1 + SIFT3D_MAX_FRAMES, 2};
    if (knob < 0 || knob >= SIFT3D_TUNE_COUNT || value < lo[knob] || value > hi[knob])
        return set_err(c, SIFT3D_ERR_ARG, "sift3d_set_tuning: knob %d does not take %d", knob, value);
    c->tune[knob] = value;
    return SIFT3D_OK;
}

extern "C" int64_t sift3d_host_buffer_grows(const sift3d_ctx *c) { return c ? c->host_grows : 0; }

#ifdef SIFT3D_DEV
extern "C" int sift3d_dev_set_stop(sift3d_ctx *c, int n)
{
    if (!c) return SIFT3D_ERR_ARG;
    c->dev_stop = n;
    return SIFT3D_OK;
}
#endif

extern "C" int sift3d_set_stream(sift3d_ctx *c, void *s)
{
    if (!c) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (s) {
        if (c->own_stream) hipStreamDestroy(c->stream);
        c->stream = (hipStream_t)s;
        c->own_stream = false;
    } else if (!c->own_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return SIFT3D_OK;
}

extern "C" int sift3d_set_max_octaves(sift3d_ctx *c, int n)
{
    if (!c || n < 0) return c ? set_err(c, SIFT3D_ERR_ARG, "max_octaves must be >= 0") : SIFT3D_ERR_ARG;
    c->max_octaves = n;
    return SIFT3D_OK;
}

extern "C" int sift3d_sync(sift3d_ctx *c)
{
    if (!c) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SIFT3D_OK;
}

/* ---- self-test: LDS float atomic add == vector ALU add ------------------- */
__global__ void selftest_lds_add_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n,
                                        float *__restrict__ valu, float *__restrict__ lds)
{
    __shared__ float cell[256];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const float x = i < n ? a[i] : 0.0f, y = i < n ? b[i] : 0.0f;
    cell[threadIdx.x] = x;
    __syncthreads();
    __hip_atomic_fetch_add(&cell[threadIdx.x], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (i < n) {
        valu[i] = x + y;
        lds[i] = cell[threadIdx.x];
    }
}

extern "C" int sift3d_selftest_lds_add(sift3d_ctx *c, const float *a, const float *b, int64_t n, float *valu, float *lds)
{
    NEED_LEVELS(c);
    if (!c || !a || !b || !valu || !lds || n <= 0 || 4 * n > c->capTot) return c ? set_err(c, SIFT3D_ERR_ARG, "bad self-test arguments") : SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    float *da = c->L[0], *db = c->L[0] + n, *dv = c->L[0] + 2 * n, *dl = c->L[0] + 3 * n;
    HIPCHK(c, hipMemcpyAsync(da, a, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(db, b, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(selftest_lds_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, da, db, (long long)n, dv, dl);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(valu, dv, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(lds, dl, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_volume = false;
    c->pad_nx = 0; /* the level buffers were used as dense scratch: their pad columns must be cleared again */
    return SIFT3D_OK;
}

/* ---- timing ------------------------------------------------------------ */
static hipEvent_t get_event(sift3d_ctx *c)
{
    if (c->pool_used == c->pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        c->pool.push_back(e);
    }
    return c->pool[c->pool_used++];
}

struct stage_scope {
    sift3d_ctx *c;
    int stage;
    hipEvent_t e0, e1;
    int ntaps;
    int64_t nvox;
    double bytes;
    hipStream_t st;
    bool timed;
    stage_scope(sift3d_ctx *c_, int stage_, double bytes_, int ntaps_ = 0, int64_t nvox_ = 0, hipStream_t st_ = nullptr)
        : c(c_), stage(stage_), e0(nullptr), e1(nullptr), ntaps(ntaps_), nvox(nvox_), bytes(bytes_), st(st_ ? st_ : c_->stream)
    {
        c->last.launches[stage] += 1;
        c->last.alg_bytes[stage] += bytes;
        /* events cost a few microseconds each (two per launch, ~170 launches: 1 ms of a 13 ms run at 512^3): mode 2
         * keeps them to the dominant kernels, the blur launches on the full-size volume */
        timed = c->timing == 1 || c->timing == 3 ||
                (c->timing == 2 && nvox == pitch_of(c->nx) * c->ny * c->nz &&
                 (stage == SIFT3D_STAGE_BLUR_FUSED || stage == SIFT3D_STAGE_BLUR_X || stage == SIFT3D_STAGE_BLUR_Y ||
                  stage == SIFT3D_STAGE_BLUR_Z_DOG));
        if (timed) {
            e0 = get_event(c);
            e1 = get_event(c);
            hipEventRecord(e0, st);
        }
    }
    void cancel() /* the launch did not happen */
    {
        c->last.launches[stage] -= 1;
        c->last.alg_bytes[stage] -= bytes;
        if (timed) c->pool_used -= 2;
        stage = -1;
    }
    ~stage_scope()
    {
        if (stage < 0) return;
        if (timed) {
            hipEventRecord(e1, st);
            c->launches.push_back({stage, e0, e1, ntaps, nvox, bytes, 0.0f, 0.0f});
        }
    }
};

static void timing_begin(sift3d_ctx *c)
{
    memset(&c->last, 0, sizeof(c->last));
    c->launches.clear();
    c->pool_used = 0;
    c->resolved = 0;
}

/* Resolves the events of every launch recorded since the last call (idempotent). */
static void timing_end(sift3d_ctx *c)
{
    if (!c->timing) return;
    hipStreamSynchronize(c->stream);
    for (size_t i = c->resolved; i < c->launches.size(); i++) {
        timed_launch &t = c->launches[i];
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) c->last.ms[t.stage] += ms;
        t.ms = ms;
        float since = 0;
        if (hipEventElapsedTime(&since, c->launches.front().e0, t.e0) != hipSuccess) since = 0;
        t.start_ms = since;
    }
    c->resolved = c->launches.size();
    if (!c->launches.empty()) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->launches.front().e0, c->launches.back().e1) == hipSuccess) c->last.total_ms = ms;
    }
}

extern "C" int sift3d_enable_timing(sift3d_ctx *c, int on)
{
    if (!c) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->timing = on < 0 ? 0 : (on > 3 ? 1 : on);
    timing_begin(c); /* operator-level *_dev calls accumulate from here until the log is read */
    return SIFT3D_OK;
}

extern "C" int sift3d_get_timings(const sift3d_ctx *c, sift3d_timings *t)
{
    if (!c || !t) return SIFT3D_ERR_ARG;
    timing_end(const_cast<sift3d_ctx *>(c));
    *t = c->last;
    return SIFT3D_OK;
}

extern "C" int sift3d_get_launch_log(const sift3d_ctx *c, sift3d_launch_record *out, int64_t cap, int64_t *n)
{
    if (!c || !n) return SIFT3D_ERR_ARG;
    timing_end(const_cast<sift3d_ctx *>(c));
    *n = (int64_t)c->launches.size();
    for (int64_t i = 0; i < *n && i < cap && out; i++) {
        const timed_launch &t = c->launches[(size_t)i];
        out[i].stage = t.stage;
        out[i].ntaps = t.ntaps;
        out[i].nvox = t.nvox;
        out[i].alg_bytes = t.bytes;
        out[i].ms = t.ms;
        out[i].start_ms = t.start_ms;
    }
    return *n > cap ? SIFT3D_ERR_CAPACITY : SIFT3D_OK;
}

/* T[0], T[1] hold at least `floats` floats each (see ctx_create).  Growing waits for the stream: a blur queued earlier may
 * still be using the old pair. */
static int ensure_T(sift3d_ctx *c, int64_t floats)
{
    if (floats <= c->capT) return SIFT3D_OK;
    if (floats > c->capN) return set_err(c, SIFT3D_ERR_ARG, "pass intermediates of %lld floats asked of a context of %lld", (long long)floats, (long long)c->capN);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(c->T[0]);
    hipFree(c->T[1]);
    c->T[0] = c->T[1] = nullptr;
    c->capT = 0;
    /* at least the second octave (capN / 8 and its row padding), so that the coarse octaves of one extraction grow it once */
    int64_t want = c->capN / 8 + c->capN / 64 + 4096;
    if (want < floats) want = floats;
    if (want > c->capN) want = c->capN;
    for (int i = 0; i < 2; i++)
        if (hipMalloc((void **)&c->T[i], sizeof(float) * (size_t)want) != hipSuccess) {
            (void)hipGetLastError();
            return set_err(c, SIFT3D_ERR_MEMORY, "out of device memory for the pass intermediates (%lld floats)", (long long)want);
        }
    c->capT = want;
    return SIFT3D_OK;
}

/* ---- device-level building blocks -------------------------------------- */
/* out = blur(in); if dog != NULL also dog = in - out.  out may be NULL when only the DoG is wanted.  Uses T[0], T[1]. */
static int blur_dev(sift3d_ctx *c, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, float sigma,
                    float min_value)
{
    float taps[SIFT3D_MAX_TAPS];
    hipStream_t ws = c->stream;
    int n = sift3d_gauss_taps(sigma, min_value, taps);
    if (n < 0) return set_err(c, SIFT3D_ERR_ARG, "bad blur parameters sigma=%g min=%g", sigma, min_value);
    const double N = (double)X * Y * Z;
    if (n == 1) { /* delta filter: out = 1*in */
        if (out) HIPCHK(c, hipMemcpyAsync(out, in, sizeof(float) * (size_t)N, hipMemcpyDeviceToDevice, ws));
        if (dog) HIPCHK(c, hipMemsetAsync(dog, 0, sizeof(float) * (size_t)N, ws));
        return SIFT3D_OK;
    }
    if (n / 2 > SIFT3D_FAST_MAX_R)
        HIPCHK(c, hipMemcpyAsync(c->d_taps, taps, sizeof(float) * n, hipMemcpyHostToDevice, ws));
    /* One fused launch per level where the volume fills the chip (it marches along z with few, fat workgroups);
     * coarse octaves keep the three-pass path.  SIFT3D_TUNE_BLUR_FUSED: 0 never / 2 always (tests, A/B timing). */
    const int fmode = c->tune[SIFT3D_TUNE_BLUR_FUSED];
    const sift3d_blur_tuning bt = {c->tune[SIFT3D_TUNE_FUSED_CHUNKS], c->tune[SIFT3D_TUNE_FUSED_ROWS], c->tune[SIFT3D_TUNE_FUSED_TILE]};
    /* measured standalone (tools/bench_blur_ab.sh 128 / 64): below 2^22 voxels the one launch still beats the three for 7 and
     * 9 taps (0.020 / 0.026 against 0.042 / 0.043 ms at 128^3), ties at 11-13 and loses at 17 */
    if (fmode == 2 || (fmode == 1 && (N >= (double)(1 << 22) || (N >= (double)(1 << 18) && n <= 9)))) {
        stage_scope sc(c, SIFT3D_STAGE_BLUR_FUSED, (dog && out ? 12.0 : 8.0) * N, n, (int64_t)N, ws);
        hipError_t e = sift3d_launch_blur_fused(ws, in, out, dog, X, Y, Z, taps, n, &bt);
        if (e == hipSuccess) return SIFT3D_OK;
        if (e != hipErrorNotSupported) HIPCHK(c, e);
        sc.cancel();
    }
    /* the three-pass form goes through the context's two intermediates: a volume beyond them (a gathered octave on a slab
     * context sized for its slab) must not overrun them */
    if ((int64_t)N > c->capN)
        return set_err(c, SIFT3D_ERR_ARG, "a %lldx%lldx%lld blur needs pass intermediates of %lld floats, the context has %lld", (long long)X,
                       (long long)Y, (long long)Z, (long long)N, (long long)c->capN);
    {
        int rc_t = ensure_T(c, (int64_t)N);
        if (rc_t) return rc_t;
    }
    float *const T0 = c->T[0], *const T1 = c->T[1];
    {
        stage_scope sc(c, SIFT3D_STAGE_BLUR_X, 8.0 * N, n, (int64_t)N, ws);
        HIPCHK(c, sift3d_launch_blur_x(ws, in, T0, X, Y, Z, taps, n, c->d_taps));
    }
    {
        stage_scope sc(c, SIFT3D_STAGE_BLUR_Y, 8.0 * N, n, (int64_t)N, ws);
        HIPCHK(c, sift3d_launch_blur_y(ws, T0, T1, X, Y, Z, taps, n, c->d_taps));
    }
    {
        stage_scope sc(c, SIFT3D_STAGE_BLUR_Z_DOG, (dog ? 16.0 : 8.0) * N, n, (int64_t)N, ws);
        HIPCHK(c, sift3d_launch_blur_z(ws, T1, out ? out : T0, dog ? in : nullptr, dog, X, Y, Z, taps, n, c->d_taps));
    }
    return SIFT3D_OK;
}

static int check_shape(sift3d_ctx *c, int64_t nx, int64_t ny, int64_t nz)
{
    if (!c) return SIFT3D_ERR_ARG;
    if (nx <= 0 || ny <= 0 || nz <= 0 || pitch_of(nx) * ny * nz > c->capN)
        return set_err(c, SIFT3D_ERR_ARG, "volume %lldx%lldx%lld does not fit the context (%lld voxels)", (long long)nx,
                       (long long)ny, (long long)nz, (long long)c->capN);
    if (nx >= (1ll << 31) || ny >= (1ll << 31) || nz >= 65536 + 2) return set_err(c, SIFT3D_ERR_ARG, "dimension too large");
    return SIFT3D_OK;
}

/* Ordering of device buffers handed to the *_dev entry points.  The context's own stream is non-blocking, i.e. not
 * ordered with the legacy default stream -- the stream the reference itself runs on, and what a caller who never
 * touched streams (torch's default stream on ROCm included) produces and consumes on.  While the context runs on its own
 * stream, every *_dev entry point therefore (in) makes its stream wait for what the default stream has queued so far and
 * (out) makes the default stream wait for what the call queued: the call behaves as if it had been issued on the default
 * stream, without a host synchronisation.  A caller that works on a stream of its own hands it over once with
 * sift3d_set_stream(); the context then runs ON that stream and no fence is needed. */
static int fence_in(sift3d_ctx *c)
{
    if (!c->own_stream) return SIFT3D_OK;
    HIPCHK(c, hipEventRecord(c->ev_fence[0], nullptr));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_fence[0], 0));
    return SIFT3D_OK;
}

static int fence_out(sift3d_ctx *c)
{
    if (!c->own_stream) return SIFT3D_OK;
    HIPCHK(c, hipEventRecord(c->ev_fence[1], c->stream));
    HIPCHK(c, hipStreamWaitEvent(nullptr, c->ev_fence[1], 0));
    return SIFT3D_OK;
}

/* The blur restricted to output planes [zo0, zo1) of the volume (the input is read as far as the filter reaches): what a
 * Z-slab rank uses to filter its two boundary bands before the interior.  Only the fused launch has that form. */
static bool blur_window_supported(int64_t X, int64_t Y, float sigma, float min_value)
{
    float taps[SIFT3D_MAX_TAPS];
    const int n = sift3d_gauss_taps(sigma, min_value, taps);
    return n >= 3 && n <= 2 * SIFT3D_FAST_MAX_R + 1 && X % 4 == 0 && X * Y < (1ll << 29);
}

static int blur_window_dev(sift3d_ctx *c, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1,
                           float sigma, float min_value)
{
    float taps[SIFT3D_MAX_TAPS];
    const int n = sift3d_gauss_taps(sigma, min_value, taps);
    if (n < 3 || zo0 < 0 || zo1 > Z || zo1 <= zo0) return set_err(c, SIFT3D_ERR_ARG, "bad blur window [%lld, %lld) of %lld planes", (long long)zo0, (long long)zo1, (long long)Z);
    const sift3d_blur_tuning bt = {c->tune[SIFT3D_TUNE_FUSED_CHUNKS], c->tune[SIFT3D_TUNE_FUSED_ROWS], c->tune[SIFT3D_TUNE_FUSED_TILE]};
    const double N = (double)X * Y * (double)(zo1 - zo0);
    stage_scope sc(c, SIFT3D_STAGE_BLUR_FUSED, (dog && out ? 12.0 : 8.0) * N, n, (int64_t)N);
    hipError_t e = sift3d_launch_blur_fused(c->stream, in, out, dog, X, Y, Z, taps, n, &bt, zo0, zo1);
    if (e == hipErrorNotSupported) {
        sc.cancel();
        return set_err(c, SIFT3D_ERR_ARG, "this shape or filter has no windowed blur (sift3d_blur_window_supported)");
    }
    HIPCHK(c, e);
    return SIFT3D_OK;
}

/* runs op between the two fences */
#define FENCED(c, op)                 \
    do {                              \
        int rc_ = fence_in(c);        \
        if (rc_) return rc_;          \
        rc_ = (op);                   \
        if (rc_) return rc_;          \
        return fence_out(c);          \
    } while (0)

/* ---- operator level ----------------------------------------------------- */
extern "C" int sift3d_gauss_blur_dev(sift3d_ctx *c, const float *d_in, float *d_out, int64_t nx, int64_t ny, int64_t nz,
                                     float sigma, float min_value)
{
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (nz < 2) return set_err(c, SIFT3D_ERR_ARG, "2-D images are outside this path (featExtract rejects z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    FENCED(c, blur_dev(c, d_in, d_out, nullptr, nx, ny, nz, sigma, min_value));
}

extern "C" int sift3d_gauss_blur_dog_dev(sift3d_ctx *c, const float *d_in, float *d_out, float *d_dog, int64_t nx,
                                         int64_t ny, int64_t nz, float sigma, float min_value)
{
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (nz < 2) return set_err(c, SIFT3D_ERR_ARG, "2-D images are outside this path (featExtract rejects z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    FENCED(c, blur_dev(c, d_in, d_out, d_dog, nx, ny, nz, sigma, min_value));
}

extern "C" int sift3d_blur_window_supported(int64_t nx, int64_t ny, float sigma, float min_value)
{
    return blur_window_supported(nx, ny, sigma, min_value) ? 1 : 0;
}

extern "C" int sift3d_gauss_blur_dog_window_dev(sift3d_ctx *c, const float *d_in, float *d_out, float *d_dog, int64_t nx, int64_t ny,
                                                int64_t nz, int64_t z_lo, int64_t z_hi, float sigma, float min_value)
{
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (nz < 2 || (!d_out && !d_dog)) return set_err(c, SIFT3D_ERR_ARG, "bad windowed blur arguments");
    HIPCHK(c, hipSetDevice(c->device));
    FENCED(c, blur_window_dev(c, d_in, d_out, d_dog, nx, ny, nz, z_lo, z_hi, sigma, min_value));
}

extern "C" int sift3d_gauss_blur(sift3d_ctx *c, const float *in, float *out, int64_t nx, int64_t ny, int64_t nz,
                                 float sigma, float min_value)
{
    NEED_LEVELS(c);
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!in || !out) return set_err(c, SIFT3D_ERR_ARG, "null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t b = sizeof(float) * (size_t)(nx * ny * nz);
    HIPCHK(c, hipMemcpyAsync(c->vol, in, b, hipMemcpyHostToDevice, c->stream));
    rc = sift3d_gauss_blur_dev(c, c->vol, c->L[0], nx, ny, nz, sigma, min_value);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->L[0], b, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_volume = false;
    c->pad_nx = 0; /* the level buffers were used as dense scratch: their pad columns must be cleared again */
    return SIFT3D_OK;
}

extern "C" int sift3d_dog_dev(sift3d_ctx *c, const float *d_a, const float *d_b, float *d_out, int64_t n)
{
    if (!c || n <= 0) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = fence_in(c);
    if (rc) return rc;
    HIPCHK(c, sift3d_launch_dog(c->stream, d_a, d_b, d_out, n));
    return fence_out(c);
}

extern "C" int sift3d_dog(sift3d_ctx *c, const float *a, const float *b, float *out, int64_t n)
{
    NEED_LEVELS(c);
    if (!c || !a || !b || !out || n <= 0 || n > c->capN) return set_err(c, SIFT3D_ERR_ARG, "bad dog arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t by = sizeof(float) * (size_t)n;
    HIPCHK(c, hipMemcpyAsync(c->L[0], a, by, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->L[1], b, by, hipMemcpyHostToDevice, c->stream));
    int rc = sift3d_dog_dev(c, c->L[0], c->L[1], c->D[0], n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->D[0], by, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_volume = false;
    c->pad_nx = 0; /* the level buffers were used as dense scratch: their pad columns must be cleared again */
    return SIFT3D_OK;
}

extern "C" int sift3d_subsample2_dev(sift3d_ctx *c, const float *d_in, int64_t nx, int64_t ny, int64_t nz, float *d_out)
{
    if (!c || nx < 2 || ny < 2 || nz < 2) return set_err(c, SIFT3D_ERR_ARG, "subsample needs every dimension >= 2");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = fence_in(c);
    if (rc) return rc;
    HIPCHK(c, sift3d_launch_subsample(c->stream, d_in, nx, nx, ny, nz, d_out, nx / 2));
    return fence_out(c);
}

extern "C" int sift3d_subsample2(sift3d_ctx *c, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out)
{
    NEED_LEVELS(c);
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!in || !out) return set_err(c, SIFT3D_ERR_ARG, "null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->L[0], in, sizeof(float) * (size_t)(nx * ny * nz), hipMemcpyHostToDevice, c->stream));
    rc = sift3d_subsample2_dev(c, c->L[0], nx, ny, nz, c->L[1]);
    if (rc) return rc;
    const size_t ob = sizeof(float) * (size_t)((nx / 2) * (ny / 2) * (nz / 2));
    HIPCHK(c, hipMemcpyAsync(out, c->L[1], ob, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_volume = false;
    c->pad_nx = 0; /* the level buffers were used as dense scratch: their pad columns must be cleared again */
    return SIFT3D_OK;
}

extern "C" int sift3d_double_size(sift3d_ctx *c, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out)
{
    NEED_LEVELS(c);
    if (!c || !in || !out || nx < 2 || ny < 2 || nz < 2 || 8 * nx * ny * nz > c->capN)
        return set_err(c, SIFT3D_ERR_ARG, "double_size: the context must hold the doubled volume");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->L[0], in, sizeof(float) * (size_t)(nx * ny * nz), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, sift3d_launch_double_size(c->stream, c->L[0], nx, ny, nz, c->L[1]));
    HIPCHK(c, hipMemcpyAsync(out, c->L[1], sizeof(float) * (size_t)(8 * nx * ny * nz), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_volume = false;
    c->pad_nx = 0; /* the level buffers were used as dense scratch: their pad columns must be cleared again */
    return SIFT3D_OK;
}

extern "C" int sift3d_halve_size(sift3d_ctx *c, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out)
{
    NEED_LEVELS(c);
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!in || !out || nx < 2 || ny < 2 || nz < 2) return set_err(c, SIFT3D_ERR_ARG, "halve_size needs every dimension >= 2");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->L[0], in, sizeof(float) * (size_t)(nx * ny * nz), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, sift3d_launch_halve_size(c->stream, c->L[0], nx, ny, nz, c->L[1]));
    const size_t ob = sizeof(float) * (size_t)((nx / 2) * (ny / 2) * (nz / 2));
    HIPCHK(c, hipMemcpyAsync(out, c->L[1], ob, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_volume = false;
    c->pad_nx = 0; /* the level buffers were used as dense scratch: their pad columns must be cleared again */
    return SIFT3D_OK;
}

/* Extrema of all levels go to one (key, value) buffer: reset, any number of appends (one kernel
 * launch each, nothing synchronises), then finalize = one host synchronisation for the count, a
 * replay of the recorded launches into a bigger buffer if it overflowed (the DoG levels stay
 * resident), and the device radix sort. */
/* on: the stream the clears are queued on (the caller orders the extrema passes behind it) */
static int cand_reset(sift3d_ctx *c, hipStream_t on = nullptr)
{
    if (!on) on = c->stream;
    c->jobs.clear();
    HIPCHK(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned long long) * 4, on));
    /* every extrema pass of the run gets its own counter set: one memset here instead of one per pass */
    HIPCHK(c, hipMemsetAsync(c->surv_counts, 0, sizeof(unsigned long long) * SIFT3D_SURV_COUNTERS * SIFT3D_SURV_SETS, on));
    HIPCHK(c, hipMemsetAsync(c->list2_counts, 0, sizeof(unsigned long long) * SIFT3D_LIST2_COUNTERS * SIFT3D_SURV_SETS, on));
    c->surv_set = 0;
    return SIFT3D_OK;
}

static int cand_append(sift3d_ctx *c, const level_job &j, bool record)
{
    c->count_queued = false;
    if (record) c->jobs.push_back(j);
    hipStream_t st = c->cand_stream ? c->cand_stream : c->stream;
    stage_scope sc(c, SIFT3D_STAGE_EXTREMA, 4.0 * (double)j.X * j.Y * j.Z, 0, j.X * j.Y * j.Z, st);
    /* own-level extrema are ~0.3 % of the voxels on blob fields (7 % on white noise): the list of a level is
     * sized at 1/surv_div of its voxels; an overflow is flagged on the device and handled in cand_finalize */
    int64_t cover = j.X * j.Y * j.Z / c->surv_div + 64 * 1024; /* split evenly over 64 segments */
    if (cover > c->surv_cap) cover = c->surv_cap;
    sift3d_survivor *surv = c->surv;
    if (c->surv_sel > 0) { /* a pass on the second extrema stream: that stream's own list, grown on demand */
        if (c->surv2_cap < cover) {
            HIPCHK(c, hipStreamSynchronize(st));
            hipFree(c->surv2);
            c->surv2 = nullptr;
            c->surv2_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->surv2, sizeof(sift3d_survivor) * (size_t)cover));
            c->surv2_cap = cover;
        }
        surv = c->surv2;
    }
    const bool fresh = c->surv_set < SIFT3D_SURV_SETS;
    const int set = fresh ? c->surv_set++ : SIFT3D_SURV_SETS - 1;
    unsigned long long *counters = c->surv_counts + (size_t)set * SIFT3D_SURV_COUNTERS;
    sift3d_extrema_lazy lz;
    memset(&lz, 0, sizeof(lz));
    const bool lazy = j.prev_b || j.next_g;
    if (lazy) {
        lz.prev_b = j.prev_b;
        lz.next_g = j.next_g;
        if (j.next_g) {
            /* the second list holds a subset of the own-level list: the same capacity always suffices */
            const int li = c->surv_sel > 0 ? 1 : 0;
            if (c->list2_cap[li] < cover) {
                HIPCHK(c, hipStreamSynchronize(st)); /* an earlier pass may still be reading the list */
                hipFree(c->list2[li]);
                c->list2[li] = nullptr;
                c->list2_cap[li] = 0;
                HIPCHK(c, hipMalloc((void **)&c->list2[li], sizeof(sift3d_survivor2) * (size_t)cover));
                c->list2_cap[li] = cover;
            }
            lz.ntaps = j.next_ntaps;
            memcpy(lz.taps, j.next_taps, sizeof(lz.taps));
            lz.list2 = c->list2[li];
            lz.list2_count = c->list2_counts + (size_t)set * SIFT3D_LIST2_COUNTERS;
            lz.list2_cap = c->list2_cap[li];
            if (!fresh) HIPCHK(c, hipMemsetAsync(lz.list2_count, 0, sizeof(unsigned long long) * SIFT3D_LIST2_COUNTERS, st));
        }
    }
    HIPCHK(c, sift3d_launch_extrema(st, j.dp, j.dc, j.dn, j.X, j.Xl ? j.Xl : j.X, j.Y, j.Z, j.z_lo, j.z_hi, j.lvl_id, c->keys_a,
                                    c->vals_a, c->d_count, c->cand_cap, surv, counters, c->d_count + 2, cover, !fresh,
                                    lazy ? &lz : nullptr));
    return SIFT3D_OK;
}

static int cand_replay(sift3d_ctx *c)
{
    HIPCHK(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned long long) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->surv_counts, 0, sizeof(unsigned long long) * SIFT3D_SURV_COUNTERS * SIFT3D_SURV_SETS, c->stream));
    HIPCHK(c, hipMemsetAsync(c->list2_counts, 0, sizeof(unsigned long long) * SIFT3D_LIST2_COUNTERS * SIFT3D_SURV_SETS, c->stream));
    c->surv_set = 0;
    for (const level_job &j : c->jobs) {
        int rc = cand_append(c, j, false);
        if (rc) return rc;
    }
    return SIFT3D_OK;
}

/* The count of validated extrema comes back in two steps so that a driver with several contexts can queue the read-back
 * on all of them before it waits for the first: cand_count_queue (asynchronous), cand_finalize (waits, replays the extrema
 * launches into bigger lists if one overflowed, sorts). */
static int cand_count_queue(sift3d_ctx *c)
{
    unsigned long long *cnt = c->h_cnt0 + 4; /* validated extrema, survivors of the last level, survivor overflow */
    cnt[0] = cnt[1] = cnt[2] = 0;
    HIPCHK(c, hipMemcpyAsync(cnt, c->d_count, sizeof(unsigned long long) * 3, hipMemcpyDeviceToHost, c->stream));
    c->count_queued = true;
    return SIFT3D_OK;
}

static int cand_finalize(sift3d_ctx *c, int64_t *count_out)
{
    for (int attempt = 0; attempt < 4; attempt++) {
        const unsigned long long *cnt = c->h_cnt0 + 4;
        if (!c->count_queued) {
            int rc = cand_count_queue(c);
            if (rc) return rc;
        }
        c->count_queued = false;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (cnt[2] > 0) { /* an own-level list was cut short: make room and redo the extrema launches */
            c->surv_div = 1;
            if ((int64_t)cnt[2] > c->surv_cap) {
                hipFree(c->surv);
                c->surv = nullptr;
                c->surv_cap = (int64_t)cnt[2] + (int64_t)cnt[2] / 4 + 4096;
                HIPCHK(c, hipMalloc((void **)&c->surv, sizeof(sift3d_survivor) * (size_t)c->surv_cap));
            }
            int rc = cand_replay(c);
            if (rc) return rc;
            continue;
        }
        if ((int64_t)cnt[0] > c->cand_cap) {
            if (alloc_cands(c, (int64_t)cnt[0] + (int64_t)cnt[0] / 4 + 4096) != SIFT3D_OK)
                return set_err(c, SIFT3D_ERR_MEMORY, "candidate buffer could not be grown to %llu entries", cnt[0]);
            int rc = cand_replay(c);
            if (rc) return rc;
            continue;
        }
        if (cnt[0] > 0)
            HIPCHK(c, sift3d_sort_candidates(c->stream, c->sort_tmp, c->sort_tmp_bytes, c->keys_a, c->keys_b, c->vals_a, c->vals_b, (int64_t)cnt[0]));
        *count_out = (int64_t)cnt[0];
        return SIFT3D_OK;
    }
    return set_err(c, SIFT3D_ERR_MEMORY, "extrema buffers could not be grown");
}

extern "C" int sift3d_extrema(sift3d_ctx *c, const float *d_prev, const float *d_cur, const float *d_next, int64_t nx,
                              int64_t ny, int64_t nz, sift3d_extremum *minima, int64_t cap_min, int64_t *n_min,
                              sift3d_extremum *maxima, int64_t cap_max, int64_t *n_max)
{
    NEED_LEVELS(c);
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!d_prev || !d_cur || !n_min || !n_max) return set_err(c, SIFT3D_ERR_ARG, "null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t b = sizeof(float) * (size_t)(nx * ny * nz);
    HIPCHK(c, hipMemcpyAsync(c->D[0], d_prev, b, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->D[1], d_cur, b, hipMemcpyHostToDevice, c->stream));
    if (d_next) HIPCHK(c, hipMemcpyAsync(c->D[2], d_next, b, hipMemcpyHostToDevice, c->stream));
    c->has_volume = false;
    c->pad_nx = 0; /* the level buffers were used as dense scratch: their pad columns must be cleared again */
    int64_t cnt = 0;
    rc = cand_reset(c);
    if (!rc) rc = cand_append(c, {c->D[0], c->D[1], d_next ? c->D[2] : nullptr, nx, ny, nz, 0, (int)nz, 0}, true);
    if (!rc) rc = cand_finalize(c, &cnt);
    if (rc) return rc;
    std::vector<unsigned long long> keys((size_t)cnt);
    std::vector<sift3d_cval> vals((size_t)cnt);
    if (cnt) {
        HIPCHK(c, hipMemcpyAsync(keys.data(), c->keys_b, sizeof(unsigned long long) * (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(vals.data(), c->vals_b, sizeof(sift3d_cval) * (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int64_t a = 0, m = 0;
    bool over = false;
    for (int64_t i = 0; i < cnt; i++) {
        const int64_t idx = (int64_t)(keys[(size_t)i] & SIFT3D_KEY_IDX_MASK);
        sift3d_extremum e;
        e.x = (int32_t)(idx % nx);
        e.y = (int32_t)((idx / nx) % ny);
        e.z = (int32_t)(idx / (nx * ny));
        e.value = vals[(size_t)i].value;
        if ((keys[(size_t)i] >> SIFT3D_KEY_MAX_SHIFT) & 1ull) {
            if (m < cap_max && maxima) maxima[m] = e; else over = true;
            m++;
        } else {
            if (a < cap_min && minima) minima[a] = e; else over = true;
            a++;
        }
    }
    *n_min = a;
    *n_max = m;
    return over ? set_err(c, SIFT3D_ERR_CAPACITY, "extrema lists need %lld + %lld entries", (long long)a, (long long)m)
                : SIFT3D_OK;
}

/* ---- pipeline ------------------------------------------------------------ */
/* A level buffer the default pipeline does not need (D[4]): allocated, and cleared like the others, the first time an
 * octave has to store that level in full. */
static int ensure_level_buffer(sift3d_ctx *c, float **buf)
{
    if (*buf) return SIFT3D_OK;
    if (hipMalloc((void **)buf, sizeof(float) * (size_t)c->capTot) != hipSuccess) {
        *buf = nullptr;
        return set_err(c, SIFT3D_ERR_MEMORY, "a level buffer of %lld floats could not be allocated", (long long)c->capTot);
    }
    HIPCHK(c, hipMemsetAsync(*buf, 0, sizeof(float) * (size_t)c->capTot, c->stream));
    return SIFT3D_OK;
}

/* The pipeline's copy of the volume has its rows padded to whole 16-byte vectors (octave_list).  When the padded
 * geometry changes, every level buffer is cleared once: the pad columns are never written afterwards. */
static int load_volume(sift3d_ctx *c, const float *src, bool from_host, int64_t nx, int64_t ny, int64_t nz)
{
    const int64_t xp = pitch_of(nx);
    if (xp != nx && (c->pad_nx != nx || c->pad_ny != ny || c->pad_nz != nz)) {
        for (int i = 0; i < 6; i++)
            if (c->L[i]) HIPCHK(c, hipMemsetAsync(c->L[i], 0, sizeof(float) * (size_t)c->capTot, c->stream));
        for (int i = 0; i < 5; i++)
            if (c->D[i]) HIPCHK(c, hipMemsetAsync(c->D[i], 0, sizeof(float) * (size_t)c->capTot, c->stream));
        HIPCHK(c, hipMemsetAsync(c->D4tiny, 0, sizeof(float) * SIFT3D_D4TINY_FLOATS, c->stream));
        HIPCHK(c, hipMemsetAsync(c->vol, 0, sizeof(float) * (size_t)c->capN, c->stream));
        c->pad_nx = nx; c->pad_ny = ny; c->pad_nz = nz;
    }
    if (xp == nx) {
        c->pad_nx = 0; /* dense rows overwrite what would be pad columns of another geometry */
        if (src != c->vol)
            HIPCHK(c, hipMemcpyAsync(c->vol, src, sizeof(float) * (size_t)(nx * ny * nz), from_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
    } else {
        if (src == c->vol) return set_err(c, SIFT3D_ERR_ARG, "in-place set_volume needs rows of whole 16-byte vectors");
        HIPCHK(c, hipMemcpy2DAsync(c->vol, sizeof(float) * (size_t)xp, src, sizeof(float) * (size_t)nx, sizeof(float) * (size_t)nx,
                                   (size_t)(ny * nz), from_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
    }
    return SIFT3D_OK;
}

extern "C" int sift3d_set_volume(sift3d_ctx *c, const float *vol, int64_t nx, int64_t ny, int64_t nz)
{
    NEED_LEVELS(c);
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!vol) return set_err(c, SIFT3D_ERR_ARG, "null volume");
    if (nz <= 1) return set_err(c, SIFT3D_ERR_ARG, "Could not read volume (z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    rc = load_volume(c, vol, true, nx, ny, nz);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nx = nx; c->ny = ny; c->nz = nz;
    c->has_volume = true;
    return SIFT3D_OK;
}

extern "C" int sift3d_set_volume_dev(sift3d_ctx *c, const float *d_vol, int64_t nx, int64_t ny, int64_t nz)
{
    NEED_LEVELS(c);
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!d_vol) return set_err(c, SIFT3D_ERR_ARG, "null volume");
    if (nz <= 1) return set_err(c, SIFT3D_ERR_ARG, "Could not read volume (z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    rc = fence_in(c); /* the caller's volume may still be in the making on the default stream */
    if (rc) return rc;
    rc = load_volume(c, d_vol, false, nx, ny, nz);
    if (rc) return rc;
    c->nx = nx; c->ny = ny; c->nz = nz;
    c->has_volume = true;
    return fence_out(c); /* and the caller may overwrite it again once the copy has run */
}

extern "C" int sift3d_set_volume_resized(sift3d_ctx *c, const float *vol, int64_t nx, int64_t ny, int64_t nz, int resize)
{
    NEED_LEVELS(c);
    if (resize == 0) return sift3d_set_volume(c, vol, nx, ny, nz);
    if (!c || !vol || nx < 2 || ny < 2 || nz < 2 || nx * ny * nz > c->capN)
        return set_err(c, SIFT3D_ERR_ARG, "set_volume_resized: bad shape or null volume");
    const int64_t ox = resize > 0 ? 2 * nx : nx / 2, oy = resize > 0 ? 2 * ny : ny / 2, oz = resize > 0 ? 2 * nz : nz / 2;
    int rc = check_shape(c, ox, oy, oz); /* the doubled volume must fit the context */
    if (rc) return rc;
    if (oz <= 1) return set_err(c, SIFT3D_ERR_ARG, "Could not read volume (z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    /* T[0], T[1]: the dense scratch volumes of the three-pass blur, free until the pyramid runs */
    rc = ensure_T(c, ox * oy * oz > nx * ny * nz ? ox * oy * oz : nx * ny * nz);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->T[0], vol, sizeof(float) * (size_t)(nx * ny * nz), hipMemcpyHostToDevice, c->stream));
    if (resize > 0) HIPCHK(c, sift3d_launch_double_size(c->stream, c->T[0], nx, ny, nz, c->T[1]));
    else HIPCHK(c, sift3d_launch_halve_size(c->stream, c->T[0], nx, ny, nz, c->T[1]));
    rc = load_volume(c, c->T[1], false, ox, oy, oz);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* the caller may free vol */
    c->nx = ox; c->ny = oy; c->nz = oz;
    c->has_volume = true;
    return SIFT3D_OK;
}

/* The pinned host buffers the descriptor kernel stores its records into, for at least `need` records.  Growing keeps the
 * first `keep` records (those of chunks already launched).  The caller has made sure nothing is writing into them. */
static int ensure_host_records(sift3d_ctx *c, int64_t need, int64_t keep)
{
    if (need <= c->hrecs_cap) return SIFT3D_OK;
    sift3d_feature *nr = nullptr;
    int *ng = nullptr;
    const int64_t cap = need + need / 8 + 1024;
    HIPCHK(c, hipHostMalloc((void **)&nr, sizeof(sift3d_feature) * (size_t)cap, hipHostMallocDefault));
    if (hipHostMalloc((void **)&ng, sizeof(int) * (size_t)cap, hipHostMallocDefault) != hipSuccess) {
        hipHostFree(nr);
        return set_err(c, SIFT3D_ERR_MEMORY, "out of pinned host memory for %lld records", (long long)cap);
    }
    if (keep > 0 && c->h_recs) {
        memcpy(nr, c->h_recs, sizeof(sift3d_feature) * (size_t)keep);
        memcpy(ng, c->h_group, sizeof(int) * (size_t)keep);
    }
    if (c->h_recs) hipHostFree(c->h_recs);
    if (c->h_group) hipHostFree(c->h_group);
    c->h_recs = nr;
    c->h_group = ng;
    c->hrecs_cap = cap;
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->d_hrecs, c->h_recs, 0));
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->d_hgroup, c->h_group, 0));
    return SIFT3D_OK;
}

/* Buffers of the per-keypoint stage for ncand candidates.  A candidate yields at most 1 + SIFT3D_MAX_FRAMES records
 * (determineCanonicalOrientation3D stops at that many frames), 4.2 on average on blob fields.  The device-side record map
 * (8 bytes a slot) is sized for the worst case so that the chunks of the stage can write it before the host knows a total;
 * the pinned host buffers (328 bytes a record) are sized for SIFT3D_TUNE_HOST_RECORDS records per candidate (default 5) and
 * grown by describe_launch when a run turns out to need more -- the worst case would be 12 records per candidate of
 * page-locked memory, tens of GB on an extrema-dense volume (advisor, round 3). */
static int ensure_kp_buffers(sift3d_ctx *c, int64_t ncand)
{
    if (ncand > c->kps_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->kp_stream));
        hipFree(c->kps); hipFree(c->nrec); hipFree(c->offs); hipFree(c->scan_tmp); hipFree(c->patch0);
        hipFree(c->rec_kp); hipFree(c->rec_frame);
        c->kps = nullptr;
        c->patch0 = nullptr;
        c->nrec = c->offs = c->rec_kp = c->rec_frame = nullptr;
        c->scan_tmp = nullptr;
        c->kps_cap = 0;
        const int64_t cap = ncand + ncand / 2 + 1024, rcap = cap * (1 + SIFT3D_MAX_FRAMES);
        c->scan_tmp_bytes = sift3d_scan_temp_bytes(cap) + 256;
        HIPCHK(c, hipMalloc((void **)&c->kps, sizeof(sift3d_dkp) * (size_t)cap));
        HIPCHK(c, hipMalloc((void **)&c->patch0, sizeof(float) * SIFT3D_PATCH_VOX * (size_t)cap));
        HIPCHK(c, hipMalloc((void **)&c->nrec, sizeof(int) * (size_t)cap));
        HIPCHK(c, hipMalloc((void **)&c->offs, sizeof(int) * (size_t)cap));
        HIPCHK(c, hipMalloc(&c->scan_tmp, c->scan_tmp_bytes));
        HIPCHK(c, hipMalloc((void **)&c->rec_kp, sizeof(int) * (size_t)rcap));
        HIPCHK(c, hipMalloc((void **)&c->rec_frame, sizeof(int) * (size_t)rcap));
        c->kps_cap = cap;
        c->recs_cap = rcap;
    }
    int per = c->tune[SIFT3D_TUNE_HOST_RECORDS];
    if (per < 1) per = 1;
    if (per > 1 + SIFT3D_MAX_FRAMES) per = 1 + SIFT3D_MAX_FRAMES;
    if (c->hrecs_cap < ncand * per) { /* nothing of an earlier run is in flight here: describe_finish has synchronised */
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->kp_stream));
        int rc = ensure_host_records(c, ncand * per, 0);
        if (rc) return rc;
    }
    return SIFT3D_OK;
}

/* Sorted candidates -> host list with whole-volume coordinates (sift3d_detect, slab tests). */
static int candidates_to_host(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand,
                              sift3d_candidate **cands_out, int64_t *n_out)
{
    std::vector<unsigned long long> keys((size_t)ncand);
    std::vector<sift3d_cval> vals((size_t)ncand);
    if (ncand) {
        HIPCHK(c, hipMemcpyAsync(keys.data(), c->keys_b, sizeof(unsigned long long) * (size_t)ncand, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(vals.data(), c->vals_b, sizeof(sift3d_cval) * (size_t)ncand, hipMemcpyDeviceToHost, c->stream));
    }
    timing_end(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    sift3d_candidate *out = (sift3d_candidate *)malloc(sizeof(sift3d_candidate) * (size_t)(ncand ? ncand : 1));
    if (!out) return set_err(c, SIFT3D_ERR_MEMORY, "out of host memory");
    for (int64_t i = 0; i < ncand; i++) {
        const unsigned long long k = keys[(size_t)i];
        const int id = (int)(k >> SIFT3D_KEY_LVL_SHIFT);
        const int64_t idx = (int64_t)(k & SIFT3D_KEY_IDX_MASK);
        if (id < 0 || id >= (int)levels.size()) {
            free(out);
            return set_err(c, SIFT3D_ERR_ARG, "candidate with level id %d outside the level table", id);
        }
        const sift3d_level &lv = levels[(size_t)id];
        sift3d_candidate &o = out[i];
        o.octave = id / 3;
        o.level = id % 3 + 1;
        o.is_max = (int)((k >> SIFT3D_KEY_MAX_SHIFT) & 1ull);
        o.x = (int32_t)(idx % lv.XP);
        o.y = (int32_t)((idx / lv.XP) % lv.Y);
        o.z = (int32_t)(idx / ((int64_t)lv.XP * lv.Y)) + lv.z_off;
        o.value = vals[(size_t)i].value;
        o.h_value = vals[(size_t)i].h;
        o.l_value = vals[(size_t)i].l;
    }
    *cands_out = out;
    *n_out = ncand;
    return SIFT3D_OK;
}

/* ---- per-keypoint stage: sorted candidates -> records in the pinned host buffer ------------------------------------
 * The two kernels of the stage are bound by different units of a CU (the keypoint kernel by its LDS
 * atomics, the descriptor kernel by the L1's miss handling), so the sorted list is cut into chunks and the keypoint kernel
 * of chunk i+1 runs on the main stream beside the descriptor kernel of chunk i on a second one.  Per chunk, on the main
 * stream: keypoint kernel -> scan of the record counts inside the chunk -> record map (which also leaves the chunk's
 * first-record index for the next chunk on the device) -> read-back of the chunk's end index.  The host then walks the
 * chunks: wait for chunk i's end index (the device is already busy with chunk i+1), launch its descriptor kernel with an
 * exact grid on the second stream.  The records are stored straight into pinned host memory while the kernels run: 60 MB
 * cross the bus beside the compute at 512^3 and nothing is left to copy at the end.
 * Three phases so that a driver with several contexts (the Z-slab driver) can keep all its devices busy:
 *   describe_queue   everything on the main stream for all chunks (returns at once)
 *   describe_launch  the descriptor launches (waits, chunk by chunk, for the keypoint side)
 *   describe_finish  the one synchronisation at the end. */
static void kp_params_of(sift3d_ctx *c, int desc_mode, float eig_thres, float size_factor, sift3d_kp_params &p)
{
    p.levels = c->d_levels;
    p.eig_thres = eig_thres;
    p.size_factor = size_factor;
    p.desc_mode = desc_mode;
    p.debug_stop = c->dev_stop;
    p.patch0 = c->patch0;
    /* workgroups per CU in the descriptor kernel's sampling phase at a time: by measurement at 512^3 (descriptor kernel 4.22 ms
     * without a limit; 1: 6.6, 2: 4.5, 3: 4.09, 4: 4.03, 5: 4.12, 6-12: 4.15-4.18) */
    p.sampler_tokens = c->sampler_tokens;
    p.sampler_cap = c->tune[SIFT3D_TUNE_SAMPLER_CAP];
}

static int kp_chunks_for(const sift3d_ctx *c, int64_t ncand)
{
    int n = c->tune[SIFT3D_TUNE_KP_CHUNKS];
    if (n <= 0) n = SIFT3D_KP_DEFAULT_CHUNKS;
    if (n > SIFT3D_KP_MAX_CHUNKS) n = SIFT3D_KP_MAX_CHUNKS;
    if (n > ncand) n = ncand > 0 ? (int)ncand : 1;
    return n;
}

static int describe_queue(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand, int desc_mode, float eig_thres,
                          float size_factor, bool levels_on_device)
{
    float taps3[SIFT3D_MAX_TAPS];
    if (sift3d_gauss_taps(0.5f, 0.01f, taps3) != 3 || sift3d_gauss_taps((float)0.95, (float)0.01, c->kp.taps5) != 5)
        return set_err(c, SIFT3D_ERR_ARG, "unexpected patch tap counts");
    if (levels.size() > 96) return set_err(c, SIFT3D_ERR_ARG, "too many levels");
    c->kp.ncand = ncand;
    c->kp.nchunks = 0;
    c->kp.launched = 0;
    c->kp.nrec = 0;
    if (ncand <= 0) return SIFT3D_OK;
    int rc = ensure_kp_buffers(c, ncand);
    if (rc) return rc;
    if (!levels_on_device)
        HIPCHK(c, hipMemcpyAsync(c->d_levels, levels.data(), sizeof(sift3d_level) * levels.size(), hipMemcpyHostToDevice, c->stream));
    kp_params_of(c, desc_mode, eig_thres, size_factor, c->kp.p);
    const int nch = kp_chunks_for(c, ncand);
    c->kp.nchunks = nch;
    HIPCHK(c, hipMemsetAsync(c->d_count + 3, 0, sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_rec_base, 0, sizeof(int), c->stream));
    int *h_end = reinterpret_cast<int *>(c->h_cnt0 + 8); /* pinned: first record past chunk i */
    for (int i = 0; i < nch; i++) {
        const int64_t a = ncand * i / nch, b = ncand * (i + 1) / nch;
        c->kp.first[i] = a;
        c->kp.first[i + 1] = b;
        h_end[i] = 0;
        {
            sift3d_kp_params q = c->kp.p;
            q.patch0 = c->patch0 + (size_t)a * SIFT3D_PATCH_VOX; /* the kernel indexes everything by its block number */
            stage_scope sc(c, SIFT3D_STAGE_KEYPOINT, 0.0, 0, b - a);
            HIPCHK(c, sift3d_launch_keypointsA(c->stream, q, c->keys_b + a, c->vals_b + a, b - a, c->kps + a, c->nrec + a, taps3));
        }
        HIPCHK(c, sift3d_scan_counts(c->stream, c->scan_tmp, c->scan_tmp_bytes, c->nrec + a, c->offs + a, b - a));
        HIPCHK(c, sift3d_launch_recmap(c->stream, c->nrec + a, c->offs + a, b - a, (int)a, c->d_rec_base + i, c->rec_kp, c->rec_frame,
                                       c->d_count + 3));
        HIPCHK(c, hipMemcpyAsync(&h_end[i], c->d_rec_base + i + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_kpc[i], c->stream));
    }
    return SIFT3D_OK;
}

static int describe_launch(sift3d_ctx *c)
{
    const int nch = c->kp.nchunks;
    if (nch <= 0) return SIFT3D_OK;
    const int *h_end = reinterpret_cast<const int *>(c->h_cnt0 + 8);
    /* one chunk: the descriptor kernel follows on the main stream; several: on the second stream, beside the next chunk's
     * keypoint kernel.  With every launch bracketed by events (timing modes 1 and 3) the launches stay on the main stream,
     * so that an event pair times its kernel alone. */
    const bool beside = nch > 1 && !(c->timing == 1 || c->timing == 3);
    hipStream_t ds = beside ? c->kp_stream : c->stream;
    int64_t base = 0;
    for (int i = 0; i < nch; i++) {
        HIPCHK(c, hipEventSynchronize(c->ev_kpc[i]));
        const int64_t end = h_end[i], m = end - base;
        if (end < base || end > c->recs_cap) return set_err(c, SIFT3D_ERR_DEVICE, "record map out of range (%lld of %lld)", (long long)end, (long long)c->recs_cap);
        if (end > c->hrecs_cap) {
            /* more records than the pinned buffers were sized for: wait for the descriptor launches of the earlier chunks
             * (they store into the buffers about to be replaced), grow, carry their records over */
            HIPCHK(c, hipStreamSynchronize(ds));
            /* chunks still to come: assume they yield records at the rate seen so far */
            const int64_t done_cand = c->kp.first[i + 1], need = done_cand > 0 && i + 1 < nch ? (int64_t)((double)end * (double)c->kp.ncand / (double)done_cand) + 1 : end;
            int rc = ensure_host_records(c, need > end ? need : end, base);
            if (rc) return rc;
            c->host_grows++;
        }
        if (m > 0) {
            stage_scope sc(c, SIFT3D_STAGE_DESCRIPTOR, 0.0, 0, m, ds);
            if (c->kp.p.sampler_cap > 0) /* the per-CU tokens start from zero whatever became of an earlier launch */
                HIPCHK(c, hipMemsetAsync(c->sampler_tokens, 0, sizeof(int) * SIFT3D_CU_SLOTS, ds));
            HIPCHK(c, sift3d_launch_descriptors(ds, c->kp.p, c->kps, c->rec_kp + base, c->rec_frame + base, m, c->d_hrecs + base,
                                                c->d_hgroup + base, c->kp.taps5));
        }
        base = end;
    }
    c->kp.nrec = base;
    c->kp.launched = 1;
    if (beside) { /* the main stream ends behind the descriptor launches: one synchronisation covers both */
        HIPCHK(c, hipEventRecord(c->ev_desc, c->kp_stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_desc, 0));
    }
    return SIFT3D_OK;
}

static int describe_finish(sift3d_ctx *c, int64_t *n_out)
{
    unsigned long long &nkp = c->h_cnt0[6]; /* pinned */
    nkp = 0;
    if (c->kp.nrec) HIPCHK(c, hipMemcpyAsync(&nkp, c->d_count + 3, sizeof(nkp), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_end(c);
    c->last.n_records = c->kp.nrec;
    c->last.n_keypoints = (int64_t)nkp;
    *n_out = c->kp.nrec;
    return SIFT3D_OK;
}

static int describe_sorted(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand, int desc_mode,
                           float eig_thres, float size_factor, int64_t *n_out, bool levels_on_device = false)
{
    int rc = describe_queue(c, levels, ncand, desc_mode, eig_thres, size_factor, levels_on_device);
    if (!rc) rc = describe_launch(c);
    if (!rc) rc = describe_finish(c, n_out);
    return rc;
}

/* The whole single-GPU path.  Host synchronisations: the extrema count, the record count, the
 * final download -- everything else is queued on the stream. */
static int run_pipeline(sift3d_ctx *c, float init_scale, bool extract, int desc_mode, float eig_thres, float size_factor,
                        sift3d_candidate **cands_out, sift3d_feature **feats_out, int64_t *n_out)
{
    if (!c) return SIFT3D_ERR_ARG;
    if (c->lean) return set_err(c, SIFT3D_ERR_ARG, "a slab context holds no pyramid (sift3d_create_slab)");
    if (!c->has_volume) return set_err(c, SIFT3D_ERR_ARG, "no volume set (sift3d_set_volume)");
    HIPCHK(c, hipSetDevice(c->device));
    timing_begin(c);
    std::vector<octave_dims> oct = octave_list(c->nx, c->ny, c->nz);
    if (c->max_octaves > 0 && oct.size() > (size_t)c->max_octaves) oct.resize((size_t)c->max_octaves);

    /* sigma schedule, MultiScale.cpp:288-294,369,526-527 (float arithmetic as there) */
    float sigma_init = 0.5f;
    if (init_scale > 0) sigma_init /= init_scale;
    float sigma = 1.6f;
    const float factor = (float)pow(2.0, 1.0 / (double)3);
    const float extra0 = sqrtf(sigma * sigma - sigma_init * sigma_init);

    const int64_t xp0 = pitch_of(c->nx);
    int rc = blur_dev(c, c->vol, c->L[0], nullptr, xp0, c->ny, c->nz, extra0, 0.01f);
    if (rc) return rc;
    if (xp0 != c->nx) HIPCHK(c, sift3d_launch_zero_pad(c->stream, c->L[0], nullptr, xp0, c->nx, c->ny * c->nz));
    /* the counters of the extrema passes are cleared on the first extrema stream, idle until octave 1's levels are done,
     * instead of between two blur launches of the main one (1.5 MB of counters: 25 us); the other streams that run
     * extrema passes wait for ev_reset */
    rc = cand_reset(c, c->ex_stream);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_reset, c->ex_stream));

    int64_t tiny_base = -1; /* float offset of the first octave of at most SIFT3D_TINY_VOX voxels */
    for (const octave_dims &d : oct)
        if (tiny_base < 0 && d.X * d.Y * d.Z <= SIFT3D_TINY_VOX) tiny_base = d.off;
    std::vector<sift3d_level> levels(oct.size() * 3);
    float fscale = 1;
    float sig[7];
    struct ex_plan {
        bool tiny_done, lazy, lazy_next;
        float *d4tiny;
        int next_ntaps;
        float next_taps[2 * SIFT3D_FAST_MAX_R + 1];
        float sig[7];
        float fscale;
    };
    std::vector<ex_plan> plans(oct.size());
    bool used_second = false;
    /* One chain of levels on the main stream.  (Round 3 tried two: the octaves after the first -- some sixty small launches
     * bound by launch latency, 0.5 ms of kernel time -- on a stream of their own from the moment the second octave's level 0
     * exists, beside the first octave's last level and its extrema passes.  It cannot overlap: the fused blur runs one
     * 84 KB-LDS workgroup per CU and two of them never share one, and the extrema march holds 8 wavefronts x 234 registers
     * per CU, so the chain's first launches wait for the grid in front of them to drain either way -- 10.45 ms per 512^3
     * extraction against 10.11; on a high-priority stream every launch of the chain took 55 - 120 us: 11.6 ms.) */
    /* the three detection levels of octave o on extrema stream `which` (0: ex_stream, 1: ex_stream2), behind everything
     * queued on the main stream so far */
    auto enqueue_extrema = [&](size_t o, int which) -> int {
        const octave_dims &d = oct[o];
        const ex_plan &pl = plans[o];
        /* timing mode 3 (measurement only): the extrema stay on the main stream, so that every launch's event pair times
         * that launch alone instead of the launch plus whatever shares the chip with it */
        hipStream_t exs = c->timing == 3 ? c->stream : (which == 0 ? c->ex_stream : c->ex_stream2);
        if (exs != c->ex_stream) HIPCHK(c, hipStreamWaitEvent(exs, c->ev_reset, 0));
        if (exs != c->stream) {
            hipEvent_t ev = which == 0 ? c->ev_oct[0] : c->ev_ex2[0];
            HIPCHK(c, hipEventRecord(ev, c->stream));
            HIPCHK(c, hipStreamWaitEvent(exs, ev, 0));
            if (which == 1) used_second = true;
        }
        c->cand_stream = exs;
        c->surv_sel = (exs != c->stream && which == 1) ? 1 : 0;
        int rc_ = SIFT3D_OK;
        /* an octave one workgroup built whole (at most 4 096 voxels, every DoG level stored): its three detection levels in one
         * launch; the per-level jobs are recorded all the same, for a replay after a list overflow */
        const bool small_octave = pl.tiny_done && pl.d4tiny;
        if (small_octave) {
            const float *dl[5] = {c->D[0] + d.off, c->D[1] + d.off, c->D[2] + d.off, c->D[3] + d.off, pl.d4tiny};
            stage_scope sc(c, SIFT3D_STAGE_EXTREMA, 12.0 * (double)d.XP * d.Y * d.Z, 0, d.XP * d.Y * d.Z, exs);
            HIPCHK(c, sift3d_launch_extrema_octave_small(exs, dl, d.XP, d.X, d.Y, d.Z, (int)o * 3, c->keys_a, c->vals_a, c->d_count, c->cand_cap));
            c->count_queued = false;
        }
        for (int l = 0; l < 3 && !rc_; l++) {
            const int id = (int)o * 3 + l;
            const float *dnext = l < 2 ? c->D[l + 2] + d.off : (pl.tiny_done ? pl.d4tiny : (pl.lazy_next ? nullptr : c->D[4] + d.off));
            level_job job = {c->D[l] + d.off, c->D[l + 1] + d.off, dnext, d.XP, d.Y, d.Z, 0, (int)d.Z, id, d.X};
            if (pl.lazy && l == 0) { /* the level below D_1 is L_0 - L_1 */
                job.dp = c->L[0] + d.off;
                job.prev_b = c->L[1] + d.off;
            }
            if (pl.lazy_next && l == 2) { /* the level above D_3 is L_4 - blur(L_4) */
                job.dn = nullptr;
                job.next_g = c->L[4] + d.off;
                job.next_ntaps = pl.next_ntaps;
                for (int q = 0; q < pl.next_ntaps; q++) job.next_taps[q] = pl.next_taps[q];
            }
            if (small_octave) c->jobs.push_back(job);
            else rc_ = cand_append(c, job, true);
            sift3d_level &lv = levels[(size_t)id];
            lv.img = c->L[l + 1] + d.off;
            lv.dogc = c->D[l + 1] + d.off;
            lv.X = (int)d.X; lv.Y = (int)d.Y; lv.Z = (int)d.Z;
            lv.XP = (int)d.XP;
            lv.sigma_h = pl.sig[l]; lv.sigma_c = pl.sig[l + 1]; lv.sigma_l = pl.sig[l + 2];
            lv.octave_factor = pl.fscale;
            lv.Zl = (int)d.Z;
            lv.z_off = 0;
            lv.pad = 0;
        }
        c->cand_stream = nullptr;
        c->surv_sel = 0;
        return rc_;
    };
    for (size_t o = 0; o < oct.size(); o++) {
        const octave_dims &d = oct[o];
        const double N = (double)d.X * d.Y * d.Z;
        hipStream_t ws = c->stream;
        sigma = 1.6f;
        sig[0] = sigma;
        /* an octave of at most 4096 voxels: all five levels in one single-workgroup launch instead of fifteen */
        bool tiny_done = false;
        /* the last DoG level of such an octave lives in a small buffer of its own, at the octave's offset from the first of them */
        float *const d4tiny = (tiny_base >= 0 && d.off >= tiny_base && d.off - tiny_base + d.XP * d.Y * d.Z <= SIFT3D_D4TINY_FLOATS)
                                  ? c->D4tiny + (d.off - tiny_base) : nullptr;
        if (d.X * d.Y * d.Z <= SIFT3D_TINY_VOX && d4tiny && c->tune[SIFT3D_TUNE_TINY_OCTAVE]) {
            sift3d_octave_taps ot;
            sift3d_octave_out oo;
            float sg = sigma;
            bool ok = true;
            for (int j = 1; j < 6 && ok; j++) {
                float taps[SIFT3D_MAX_TAPS];
                const int n = sift3d_gauss_taps(sg * sqrtf(factor * factor - 1.0f), 0.01f, taps);
                ok = n >= 3 && n <= 2 * SIFT3D_FAST_MAX_R + 1;
                for (int q = 0; ok && q < n; q++) ot.f[j - 1][q] = taps[q];
                ot.n[j - 1] = n;
                oo.L[j - 1] = j < 5 ? c->L[j] + d.off : nullptr;
                oo.D[j - 1] = j < 5 ? c->D[j - 1] + d.off : d4tiny;
                sg *= factor;
            }
            if (ok) {
                stage_scope sc(c, SIFT3D_STAGE_OCTAVE_TINY, 40.0 * N, 0, (int64_t)N, ws);
                hipError_t e = sift3d_launch_tiny_octave(ws, c->L[0] + d.off, oo, d.X, d.XP, d.Y, d.Z, ot);
                if (e == hipSuccess) tiny_done = true;
                else if (e != hipErrorNotSupported) HIPCHK(c, e);
                else sc.cancel();
            }
        }
        /* Levels nothing reads in full are not computed in full.  D_0 is only ever looked at around the extrema of D_1
         * and D_4 around those of D_3 (26 + 27 + 27 test), and L_5 exists only to make D_4.  The reference does the same
         * in its own way: it never materialises the DoG level above a detection level but takes G1 - G2 at the 27
         * positions (validateDifferencePeak3D, MultiScale.cpp:1135-1223) -- though it still blurs the whole volume for
         * L_5.  Here D_0 is taken as L_0 - L_1 at those positions and L_5 is filtered only in the 27-voxel neighbourhood
         * of what passed every other test (extrema_validate_lazy_kernel: same operations, same order, same bits).  Per
         * octave that is one 17-tap blur of the whole volume and two DoG stores less.  SIFT3D_TUNE_LAZY_LEVELS = 0 (A/B,
         * tests): every level stored, as before. */
        float next_taps[SIFT3D_MAX_TAPS];
        int next_ntaps = 0;
        bool lazy = !tiny_done && d.XP >= 8 && d.Y >= 3 && d.Z >= 3 && d.XP * d.Y < (1ll << 29) && c->tune[SIFT3D_TUNE_LAZY_LEVELS];
        if (lazy) {
            float sg = 1.6f; /* sigma entering j = 5, accumulated as the loop below does */
            for (int j = 1; j < 5; j++) sg *= factor;
            next_ntaps = sift3d_gauss_taps(sg * sqrtf(factor * factor - 1.0f), 0.01f, next_taps);
            if (next_ntaps != 2 * SIFT3D_FAST_MAX_R + 1) lazy = false; /* the one filter length the third phase is built for */
        }
        const bool lazy_next = lazy;
        for (int j = 1; j < 6; j++) {
            if (tiny_done) {
                if (j == 3 && o + 1 < oct.size()) {
                    stage_scope sc(c, SIFT3D_STAGE_SUBSAMPLE, 4.5 * N, 0, (int64_t)N, ws);
                    HIPCHK(c, sift3d_launch_subsample(ws, c->L[3] + d.off, d.XP, d.X, d.Y, d.Z, c->L[0] + oct[o + 1].off, oct[o + 1].XP));
                    /* the subsample writes the logical columns only: a pitched coarser octave (100 -> 50 -> pitch 52) needs its pad
                     * columns zeroed here -- the blur reads them as the zero border, and the buffer may hold an earlier volume */
                    if (oct[o + 1].XP != oct[o + 1].X)
                        HIPCHK(c, sift3d_launch_zero_pad(ws, c->L[0] + oct[o + 1].off, nullptr, oct[o + 1].XP, oct[o + 1].X, oct[o + 1].Y * oct[o + 1].Z));
                }
                sigma *= factor;
                sig[j] = sigma;
                continue;
            }
            const float ex = sigma * sqrtf(factor * factor - 1.0f);
            /* L_j = blur(L_{j-1}); D_{j-1} = L_{j-1} - L_j fused into the z pass */
            /* nothing reads L_5: only D_4 = L_4 - L_5 is needed, so the last level is not stored */
            if (!(lazy_next && j == 5)) {
                if (j == 5) {
                    rc = ensure_level_buffer(c, &c->D[4]);
                    if (rc) return rc;
                }
                float *dst_dog = (lazy && j == 1) ? nullptr : c->D[j - 1] + d.off;
                rc = blur_dev(c, c->L[j - 1] + d.off, j < 5 ? c->L[j] + d.off : nullptr, dst_dog, d.XP, d.Y, d.Z, ex, 0.01f);
                if (rc) return rc;
                if (d.XP != d.X) /* the blur ran over the pitched width: its pad columns go back to zero */
                    HIPCHK(c, sift3d_launch_zero_pad(ws, j < 5 ? c->L[j] + d.off : nullptr, dst_dog, d.XP, d.X, d.Y * d.Z));
            }
            if (j == 3 && o + 1 < oct.size()) {
                {
                    stage_scope sc(c, SIFT3D_STAGE_SUBSAMPLE, 4.5 * N, 0, (int64_t)N, ws);
                    HIPCHK(c, sift3d_launch_subsample(ws, c->L[3] + d.off, d.XP, d.X, d.Y, d.Z, c->L[0] + oct[o + 1].off, oct[o + 1].XP));
                }
                /* the subsample writes the logical columns only: a pitched coarser octave (100 -> 50 -> pitch 52) needs its pad
                 * columns zeroed here -- the blur reads them as the zero border, and the buffer may hold an earlier volume */
                if (oct[o + 1].XP != oct[o + 1].X)
                    HIPCHK(c, sift3d_launch_zero_pad(ws, c->L[0] + oct[o + 1].off, nullptr, oct[o + 1].XP, oct[o + 1].X, oct[o + 1].Y * oct[o + 1].Z));
            }
            sigma *= factor;
            sig[j] = sigma;
        }
        /* the extrema of this octave go to another stream: see enqueue_extrema above */
        ex_plan &pl = plans[o];
        pl.tiny_done = tiny_done;
        pl.d4tiny = d4tiny;
        pl.lazy = lazy;
        pl.lazy_next = lazy_next;
        pl.next_ntaps = next_ntaps;
        for (int q = 0; q < next_ntaps && q < 2 * SIFT3D_FAST_MAX_R + 1; q++) pl.next_taps[q] = next_taps[q];
        for (int q = 0; q < 7; q++) pl.sig[q] = sig[q];
        pl.fscale = fscale;
        /* Octave 0's extrema fill the chip for a millisecond, and so do octave 1's blurs for a third of one, while everything
         * coarser is a chain of small launches that leaves it idle: octave 0's extrema therefore wait until octave 1's levels
         * are done and then run beside that chain; the extrema of the coarser octaves go to a stream of their own so that
         * they do not queue up behind octave 0's.  (Started right after octave 0's own levels they shared the chip with
         * octave 1's blurs -- both three to ten times slower for it -- and the chain of octaves 2.. ran alone afterwards,
         * a millisecond of mostly idle chip: 10.75 against 10.50 ms per extraction.) */
        if (o == 0 && oct.size() == 1) {
            rc = enqueue_extrema(0, 0);
            if (rc) return rc;
        } else if (o >= 1) {
            if (o == 1) {
                rc = enqueue_extrema(0, 0);
                if (rc) return rc;
            }
            rc = enqueue_extrema(o, 1);
            if (rc) return rc;
        }
        fscale *= 2.0f;
        c->last.n_octaves++;
    }
    HIPCHK(c, hipEventRecord(c->ev_oct[1], c->ex_stream)); /* the candidate counts are read on the main stream */
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_oct[1], 0));
    if (used_second) {
        HIPCHK(c, hipEventRecord(c->ev_ex2[1], c->ex_stream2));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_ex2[1], 0));
    }
    /* the level table goes to the device now, behind the pyramid, not after the host has waited for the extrema count */
    const bool levels_early = extract && levels.size() <= 96;
    if (levels_early)
        HIPCHK(c, hipMemcpyAsync(c->d_levels, levels.data(), sizeof(sift3d_level) * levels.size(), hipMemcpyHostToDevice, c->stream));
    int64_t ncand = 0;
    rc = cand_finalize(c, &ncand);
    if (rc) return rc;
    c->last.n_extrema = ncand;
    if (!extract) return candidates_to_host(c, levels, ncand, cands_out, n_out);
    rc = describe_sorted(c, levels, ncand, desc_mode, eig_thres, size_factor, n_out, levels_early);
    if (rc) return rc;
    *feats_out = c->h_recs; /* pinned, owned by the context */
    return SIFT3D_OK;
}

/* ---- building blocks for Z-slab mode: the caller owns the level buffers (device memory), places
 * halos, and drives the exchange; the library detects and describes on whatever it is given. ---- */
extern "C" int sift3d_candidates_reset(sift3d_ctx *c)
{
    if (!c) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    timing_begin(c);
    return cand_reset(c);
}

extern "C" int sift3d_extrema_append_dev(sift3d_ctx *c, const float *d_prev, const float *d_cur, const float *d_next,
                                         int64_t nx, int64_t ny, int64_t nz_local, int level_id, int64_t z_lo, int64_t z_hi)
{
    if (!c || !d_prev || !d_cur || !d_next) return SIFT3D_ERR_ARG;
    if (nx <= 0 || ny <= 0 || nz_local <= 0 || nx >= (1ll << 31) || ny >= (1ll << 31) || nz_local >= 65538 ||
        nx * ny * nz_local > (int64_t)SIFT3D_KEY_IDX_MASK || level_id < 0 || level_id >= 96)
        return set_err(c, SIFT3D_ERR_ARG, "bad extrema_append arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = fence_in(c);
    if (rc) return rc;
    rc = cand_append(c, {d_prev, d_cur, d_next, nx, ny, nz_local, (int)z_lo, (int)z_hi, level_id}, true);
    if (rc) return rc;
    return fence_out(c); /* the caller may reuse the buffers once the pass has read them */
}

/* shapes the second and third extrema phase take neighbour levels in unstored form for */
static bool lazy_shape_ok(int64_t nx, int64_t ny, int64_t nz_local)
{
    return nx % 4 == 0 && nx >= 8 && ny >= 3 && nz_local >= 3 && nx * ny < (1ll << 29);
}

extern "C" int sift3d_lazy_levels_supported(int64_t nx, int64_t ny, int64_t nz_local, float next_sigma)
{
    float taps[SIFT3D_MAX_TAPS];
    return lazy_shape_ok(nx, ny, nz_local) && sift3d_gauss_taps(next_sigma, 0.01f, taps) == 2 * SIFT3D_FAST_MAX_R + 1 ? 1 : 0;
}

extern "C" int sift3d_extrema_append_lazy_dev(sift3d_ctx *c, const float *d_prev, const float *g_prev_a, const float *g_prev_b,
                                              const float *d_cur, const float *d_next, const float *g_next, float next_sigma,
                                              int64_t nx, int64_t ny, int64_t nz_local, int level_id, int64_t z_lo, int64_t z_hi)
{
    if (!c || !d_cur || (!d_prev && !(g_prev_a && g_prev_b)) || (!d_next && !g_next)) return SIFT3D_ERR_ARG;
    if (nx <= 0 || ny <= 0 || nz_local <= 0 || nx >= (1ll << 31) || ny >= (1ll << 31) || nz_local >= 65538 ||
        nx * ny * nz_local > (int64_t)SIFT3D_KEY_IDX_MASK || level_id < 0 || level_id >= 96)
        return set_err(c, SIFT3D_ERR_ARG, "bad extrema_append arguments");
    float taps[SIFT3D_MAX_TAPS];
    const int ntaps = d_next ? 0 : sift3d_gauss_taps(next_sigma, 0.01f, taps);
    if (((!d_prev || !d_next) && !lazy_shape_ok(nx, ny, nz_local)) || (!d_next && ntaps != 2 * SIFT3D_FAST_MAX_R + 1))
        return set_err(c, SIFT3D_ERR_ARG, "this shape or filter needs stored DoG levels (sift3d_lazy_levels_supported)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = fence_in(c);
    if (rc) return rc;
    level_job jb = {d_prev ? d_prev : g_prev_a, d_cur, d_next, nx, ny, nz_local, (int)z_lo, (int)z_hi, level_id};
    if (!d_prev) jb.prev_b = g_prev_b;
    if (!d_next) {
        jb.next_ntaps = ntaps;
        for (int q = 0; q < ntaps; q++) jb.next_taps[q] = taps[q];
        jb.next_g = g_next;
    }
    rc = cand_append(c, jb, true);
    if (rc) return rc;
    return fence_out(c);
}

static int levels_from_desc(sift3d_ctx *c, const sift3d_level_desc *ld, int n, std::vector<sift3d_level> &levels)
{
    if (!ld || n <= 0 || n > 96) return set_err(c, SIFT3D_ERR_ARG, "bad level table");
    levels.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        sift3d_level &lv = levels[(size_t)i];
        lv.img = ld[i].img;
        lv.XP = (int)ld[i].nx;
        lv.dogc = ld[i].dogc;
        lv.X = (int)ld[i].nx; lv.Y = (int)ld[i].ny; lv.Z = (int)ld[i].nz_global;
        lv.Zl = (int)ld[i].nz_local;
        lv.z_off = (int)ld[i].z_offset;
        lv.sigma_h = ld[i].sigma_h; lv.sigma_c = ld[i].sigma_c; lv.sigma_l = ld[i].sigma_l;
        lv.octave_factor = ld[i].octave_factor;
        lv.pad = 0;
    }
    return SIFT3D_OK;
}

extern "C" int sift3d_candidates_dev(sift3d_ctx *c, const sift3d_level_desc *levels, int n_levels, sift3d_candidate **out,
                                     int64_t *n_out)
{
    if (!c || !out || !n_out) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<sift3d_level> lv;
    int rc = levels_from_desc(c, levels, n_levels, lv);
    if (rc) return rc;
    rc = fence_in(c); /* a replay of the extrema passes reads the caller's level buffers again */
    if (rc) return rc;
    int64_t ncand = 0;
    rc = cand_finalize(c, &ncand);
    if (rc) return rc;
    return candidates_to_host(c, lv, ncand, out, n_out); /* ends with a host synchronisation: nothing is left in flight */
}

extern "C" int sift3d_describe_dev(sift3d_ctx *c, const sift3d_level_desc *levels, int n_levels, int desc_mode,
                                   float eig_thres, float size_factor, const sift3d_feature **view, const int32_t **group_view,
                                   int64_t *n_out)
{
    if (!c || !view || !n_out) return SIFT3D_ERR_ARG;
    if (desc_mode < SIFT3D_DESC_SIFT || desc_mode > SIFT3D_DESC_NRRIEF) return set_err(c, SIFT3D_ERR_ARG, "bad descriptor mode");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<sift3d_level> lv;
    int rc = levels_from_desc(c, levels, n_levels, lv);
    if (rc) return rc;
    rc = fence_in(c); /* the keypoint and descriptor kernels read img / dogc of the level table: the caller's buffers */
    if (rc) return rc;
    int64_t ncand = 0;
    rc = cand_finalize(c, &ncand);
    if (rc) return rc;
    c->last.n_extrema = ncand;
    rc = describe_sorted(c, lv, ncand, desc_mode, eig_thres, size_factor, n_out); /* ends with a host synchronisation */
    if (rc) return rc;
    *view = c->h_recs;
    if (group_view) *group_view = c->h_group;
    return SIFT3D_OK;
}

/* One z-slice of a resident Gaussian level of the last run, dense (nx_o * ny_o floats of octave o): what the reference's
 * debug output image.pgm shows (fioFeatureSliceXY of octave 0's first blurred level, R/src_common/MultiScale.cpp:373-384). */
extern "C" int sift3d_get_level_slice(sift3d_ctx *c, int octave, int level, int64_t z, float *out, int64_t *nx_out, int64_t *ny_out)
{
    if (!c || !out) return SIFT3D_ERR_ARG;
    NEED_LEVELS(c);
    if (!c->has_volume) return set_err(c, SIFT3D_ERR_ARG, "no volume set (sift3d_set_volume)");
    std::vector<octave_dims> oct = octave_list(c->nx, c->ny, c->nz);
    if (octave < 0 || (size_t)octave >= oct.size() || level < 0 || level > 4 || !c->L[level])
        return set_err(c, SIFT3D_ERR_ARG, "no level %d of octave %d", level, octave);
    const octave_dims &d = oct[(size_t)octave];
    if (z < 0 || z >= d.Z) return set_err(c, SIFT3D_ERR_ARG, "slice %lld outside 0..%lld", (long long)z, (long long)d.Z - 1);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy2DAsync(out, sizeof(float) * (size_t)d.X, c->L[level] + d.off + z * d.XP * d.Y, sizeof(float) * (size_t)d.XP,
                               sizeof(float) * (size_t)d.X, (size_t)d.Y, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nx_out) *nx_out = d.X;
    if (ny_out) *ny_out = d.Y;
    return SIFT3D_OK;
}

extern "C" int sift3d_detect(sift3d_ctx *c, float initial_image_scale, sift3d_candidate **out, int64_t *n_out)
{
    if (!c || !out || !n_out) return SIFT3D_ERR_ARG;
    return run_pipeline(c, initial_image_scale, false, 0, 140.0f, 1.0f, out, nullptr, n_out);
}

extern "C" int sift3d_extract_view(sift3d_ctx *c, float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                                   const sift3d_feature **view, int64_t *n_out)
{
    if (!c || !view || !n_out) return SIFT3D_ERR_ARG;
    if (desc_mode < SIFT3D_DESC_SIFT || desc_mode > SIFT3D_DESC_NRRIEF) return set_err(c, SIFT3D_ERR_ARG, "bad descriptor mode");
    sift3d_feature *v = nullptr;
    int rc = run_pipeline(c, initial_image_scale, true, desc_mode, eig_thres, size_factor, nullptr, &v, n_out);
    *view = v;
    return rc;
}

extern "C" int sift3d_extract(sift3d_ctx *c, float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                              sift3d_feature **out, int64_t *n_out)
{
    if (!c || !out || !n_out) return SIFT3D_ERR_ARG;
    const sift3d_feature *v = nullptr;
    int rc = sift3d_extract_view(c, initial_image_scale, desc_mode, eig_thres, size_factor, &v, n_out);
    if (rc) return rc;
    *out = (sift3d_feature *)malloc(sizeof(sift3d_feature) * (size_t)(*n_out ? *n_out : 1));
    if (!*out) return set_err(c, SIFT3D_ERR_MEMORY, "out of host memory");
    if (*n_out) memcpy(*out, v, sizeof(sift3d_feature) * (size_t)*n_out);
    return SIFT3D_OK;
}

/* ======================================================================================================================
 * Z-slab extraction driven from C: ONE process, one context per device, halos moved with hipMemcpyPeerAsync (xGMI between
 * the GPUs of a node).  The reference has no multi-GPU code; the partitioning is the one of
 * 3d_sift_cuda_amd/zslab.py (which runs one process per GPU over RCCL for bench.py): slabs along z with boundaries that are
 * multiples of 2^K, every level recomputed on slab +- 8 slices and its 8-slice halo refreshed from the two neighbours, the
 * other 24 slices of the L1..L3 patch halos copied once per octave on a second stream while L4, L5 and the extrema passes
 * run, the first unsharded octave assembled on rank 0.  A halo copy is queued on the RECEIVER's stream behind an event the
 * sender records when the level is complete, so no host thread ever waits inside the pyramid; the host enqueues the work
 * of all devices round-robin.  The same device may be listed several times (a rehearsal of the slab logic on one GPU).
 * ====================================================================================================================== */
namespace {
const int64_t ZS_HALO = 32; /* slices of L1..L3 kept around a slab: an 11^3 patch reaches < 29 slices from its keypoint */
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
    hipEvent_t ev_patch = nullptr;     /* the patch-halo copies into this rank are done */
    std::vector<float *> allocs;       /* what this run had to allocate beside the arena */
    float *arena = nullptr;            /* one block reused from run to run (sized after the first run of a handle) */
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


#define ZS_HIP(call)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            snprintf(errbuf, sizeof errbuf, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = SIFT3D_ERR_DEVICE;                                                                                    \
            goto done;                                                                                                 \
        }                                                                                                              \
    } while (0)
#define ZS_COMM(call)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            snprintf(errbuf, sizeof errbuf, "slab exchange: %s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = SIFT3D_ERR_COMM;                                                                                      \
            goto done;                                                                                                 \
        }                                                                                                              \
    } while (0)
#define ZS_X(call) /* a step of the exchange through the handle's transport (peer copies or RCCL) */                 \
    do {                                                                                                               \
        if ((call) != 0) {                                                                                             \
            snprintf(errbuf, sizeof errbuf, "%s", zs_transport_error(h->tr));                                          \
            rc = SIFT3D_ERR_COMM;                                                                                      \
            goto done;                                                                                                 \
        }                                                                                                              \
    } while (0)
#define ZS_RC(call)                                                                                  \
    do {                                                                                             \
        rc = (call);                                                                                 \
        if (rc != SIFT3D_OK) {                                                                       \
            snprintf(errbuf, sizeof errbuf, "rank %d: %s", r, sift3d_last_error(R[(size_t)r].c));    \
            goto done;                                                                               \
        }                                                                                            \
    } while (0)
} // namespace

struct sift3d_zslab {
    zs_plan plan;
    std::vector<int> devices;
    std::vector<zs_rank> R; /* one per slab; a single one when the volume is too thin to shard */
    int lazy_levels = 1;    /* SIFT3D_TUNE_LAZY_LEVELS */
    int bands_first = 1;    /* SIFT3D_TUNE_BANDS_FIRST */
    int transport_want = ZS_TRANSPORT_PEER; /* SIFT3D_ZSLAB_TRANSPORT */
    zs_transport *tr = nullptr;             /* created by the first extraction after the choice (zslab_transport.hip) */
    sift3d_zslab(int64_t nx, int64_t ny, int64_t nz, int n) : plan(nx, ny, nz, n) {}
};

extern "C" void sift3d_zslab_destroy(sift3d_zslab *h)
{
    if (!h) return;
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
        if (q.ev_level) hipEventDestroy(q.ev_level);
        if (q.ev_l3) hipEventDestroy(q.ev_l3);
        if (q.ev_patch) hipEventDestroy(q.ev_patch);
        sift3d_destroy(q.c);
    }
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
    h->devices.assign(devices, devices + n_devices);
    h->R.resize((size_t)S);
    for (int r = 0; r < S; r++) {
        zs_rank &q = h->R[(size_t)r];
        q.dev = devices[r];
        int64_t i0 = 0, i1 = nz;
        if (S > 1) h->plan.input_range(r, i0, i1);
        ZS_HIP(hipSetDevice(q.dev));
        /* a slab context owns no level buffers (they come from the rank's arena); its pass intermediates must hold the
         * largest volume the rank ever blurs: its slab with halos, and on rank 0 the first unsharded octave, which is
         * gathered there (nz / 2^K slices of a plane a 4^K-th the size: smaller than the slab unless the slabs are many) */
        int64_t ctx_nz = (i1 - i0) + 2 * ZS_HALO;
        if (r == 0 && S > 1 && (size_t)h->plan.K < h->plan.oct.size()) {
            const std::vector<int64_t> &g = h->plan.oct[(size_t)h->plan.K];
            const int64_t need = (pitch_of(g[0]) * g[1] * g[2] + pitch_of(nx) * ny - 1) / (pitch_of(nx) * ny);
            ctx_nz = std::max(ctx_nz, need);
        }
        q.c = ctx_create(q.dev, nx, ny, ctx_nz, S > 1);
        if (!q.c) {
            snprintf(errbuf, sizeof errbuf, "rank %d: no context on device %d (memory?)", r, q.dev);
            rc = SIFT3D_ERR_MEMORY;
            goto done;
        }
        ZS_HIP(hipStreamCreateWithFlags(&q.copy_stream, hipStreamNonBlocking));
        ZS_HIP(hipStreamCreateWithFlags(&q.halo_stream, hipStreamNonBlocking));
        ZS_HIP(hipEventCreateWithFlags(&q.ev_halo, hipEventDisableTiming));
        ZS_HIP(hipEventCreateWithFlags(&q.ev_level, hipEventDisableTiming));
        ZS_HIP(hipEventCreateWithFlags(&q.ev_l3, hipEventDisableTiming));
        ZS_HIP(hipEventCreateWithFlags(&q.ev_patch, hipEventDisableTiming));
        for (int p = 0; p < S; p++) /* direct copies between the devices where the fabric allows (errors: already on, or same device) */
            if (devices[p] != q.dev) (void)hipDeviceEnablePeerAccess(devices[p], 0);
        (void)hipGetLastError();
    }
done:
    if (status_out) *status_out = rc;
    if (rc != SIFT3D_OK) {
        sift3d_zslab_destroy(h);
        return fail(errbuf);
    }
    return h;
}

extern "C" sift3d_zslab *sift3d_zslab_create(const int *devices, int n_devices, int64_t nx, int64_t ny, int64_t nz, char *err, int64_t err_len)
{
    return zslab_create_impl(devices, n_devices, nx, ny, nz, err, err_len, nullptr);
}

extern "C" int sift3d_zslab_extract(sift3d_zslab *h, const float *vol, float initial_image_scale, int desc_mode, float eig_thres,
                                    float size_factor, sift3d_feature **out, int64_t *n_out, sift3d_zslab_stats *stats, char *err,
                                    int64_t err_len)
{
    char errbuf[512] = "";
    int rc = SIFT3D_OK;
    int r = 0;
    if (err && err_len > 0) err[0] = 0;
    if (!h || !vol || !out || !n_out || desc_mode < SIFT3D_DESC_SIFT || desc_mode > SIFT3D_DESC_NRRIEF) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "bad arguments");
        return SIFT3D_ERR_ARG;
    }
    *out = nullptr;
    *n_out = 0;
    const zs_plan &plan = h->plan;
    const int64_t nx = plan.nx, ny = plan.ny, nz = plan.nz;
    std::vector<zs_rank> &R = h->R;
    const int S = (int)R.size();
    const int K = plan.K;
    sift3d_zslab_stats st;
    memset(&st, 0, sizeof st);
    st.n_ranks = S;
    st.sharded_octaves = S > 1 ? K : 0;
    if (!h->tr) {
        std::vector<int> devs((size_t)S);
        for (int i = 0; i < S; i++) devs[(size_t)i] = R[(size_t)i].dev;
        char terr[400];
        h->tr = zs_transport_create(h->transport_want == SIFT3D_TRANSPORT_RCCL ? ZS_TRANSPORT_RCCL : ZS_TRANSPORT_PEER, devs.data(), S, terr, sizeof terr);
        if (!h->tr) {
            if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", terr);
            return SIFT3D_ERR_COMM;
        }
    }
    st.transport = zs_transport_kind(h->tr) == ZS_TRANSPORT_RCCL ? SIFT3D_TRANSPORT_RCCL : SIFT3D_TRANSPORT_PEER_COPY;
    st.transport_fell_back = zs_transport_fell_back(h->tr);
    st.rccl_version = zs_transport_version(h->tr);
    std::vector<std::vector<sift3d_feature>> recs((size_t)S);
    std::vector<std::vector<int>> grps((size_t)S);
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
    float *next0[64]; /* level 0 of the next octave per rank */
    for (int i = 0; i < 64; i++) next0[i] = nullptr;
    float fscale = 1.0f;

    /* ---- input slabs, level 0 of octave 0 ---- */
    for (r = 0; r < S; r++) {
        zs_rank &q = R[(size_t)r];
        int64_t i0 = 0, i1 = nz;
        if (S > 1) plan.input_range(r, i0, i1);
        ZS_HIP(hipSetDevice(q.dev));
        q.levels.assign(plan.oct.size() * 3, sift3d_level());
        ZS_RC(cand_reset(q.c));
        timing_begin(q.c);
        /* level 0 = initial blur of the input, on slab +- 8 from input slab +- 16 */
        const int64_t XY = nx * ny;
        float *din = q.alloc((i1 - i0) * XY), *tmp = q.alloc((i1 - i0) * XY);
        if (!din || !tmp) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "rank %d: out of device memory", r); goto done; }
        ZS_HIP(hipMemcpyAsync(din, vol + i0 * XY, sizeof(float) * (size_t)((i1 - i0) * XY), hipMemcpyHostToDevice, q.c->stream));
        ZS_RC(blur_dev(q.c, din, tmp, nullptr, nx, ny, i1 - i0, extra0, 0.01f));
        int64_t z0 = 0, z1 = nz;
        if (S > 1) plan.slab(r, 0, z0, z1);
        const bool lo = S > 1 && r > 0, hi = S > 1 && r < S - 1;
        const int64_t e0 = lo ? std::max<int64_t>(0, z0 - ZS_HALO) : z0, e1 = hi ? std::min<int64_t>(nz, z1 + ZS_HALO) : z1;
        const int64_t c0 = lo ? std::max(e0, z0 - ZS_BLUR) : e0, c1 = hi ? std::min(e1, z1 + ZS_BLUR) : e1;
        float *l0 = q.alloc((e1 - e0) * XY);
        if (!l0) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "rank %d: out of device memory", r); goto done; }
        ZS_HIP(hipMemcpyAsync(l0 + (c0 - e0) * XY, tmp + (c0 - i0) * XY, sizeof(float) * (size_t)((c1 - c0) * XY), hipMemcpyDeviceToDevice, q.c->stream));
        next0[r] = l0;
    }

    /* ---- octaves ---- */
    for (int o = 0; o < (int)plan.oct.size(); o++) {
        const int64_t X = plan.oct[(size_t)o][0], Y = plan.oct[(size_t)o][1], zo = plan.oct[(size_t)o][2], XY = X * Y;
        const bool sharded = S > 1 && o < K;
        const int nr = sharded ? S : 1; /* the gathered octaves live on rank 0 */
        /* As on one device (run_pipeline): D_0 is read as L_0 - L_1 around the extrema of D_1, and L_5 -- hence D_4 -- is
         * filtered only around the candidates of D_3, from L_4.  A slab then blurs four levels instead of five and exchanges
         * four halos per octave instead of five; the third extrema phase reads L_4 nine slices beyond a candidate, so L_4's
         * halo is refreshed nine slices deep instead of eight.  Rows that are not whole 16-byte vectors keep every level
         * stored (the extrema kernels of such rows take stored levels only), as does sift3d_zslab_set_tuning(SIFT3D_TUNE_LAZY_LEVELS, 0). */
        float taps5[SIFT3D_MAX_TAPS];
        const int ntaps5 = sift3d_gauss_taps(extras[4], 0.01f, taps5);
        const bool lazy = ntaps5 == 2 * SIFT3D_FAST_MAX_R + 1 && X % 4 == 0 && X >= 8 && XY < (1ll << 29) && Y >= 3 && zo >= 3 && h->lazy_levels;
        const int nlev = lazy ? 4 : 5;
        for (r = 0; r < nr; r++) {
            zs_rank &q = R[(size_t)r];
            if (sharded) plan.slab(r, o, q.z0, q.z1); else { q.z0 = 0; q.z1 = zo; }
            q.lo = sharded && r > 0;
            q.hi = sharded && r < S - 1;
            q.e0 = q.lo ? std::max<int64_t>(0, q.z0 - ZS_HALO) : q.z0;
            q.e1 = q.hi ? std::min<int64_t>(zo, q.z1 + ZS_HALO) : q.z1;
            ZS_HIP(hipSetDevice(q.dev));
            q.L[0] = next0[r];
            for (int j = 1; j < 6; j++) q.L[j] = j <= nlev ? q.alloc((q.e1 - q.e0) * XY) : nullptr;
            for (int j = 0; j < 5; j++) q.D[j] = (lazy && (j == 0 || j == 4)) ? nullptr : q.alloc((q.e1 - q.e0) * XY);
            for (int j = 1; j <= nlev; j++)
                if (!q.L[j] || (!q.D[j - 1] && !(lazy && j == 1))) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "rank %d: out of device memory", r); goto done; }
        }
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
            for (r = 0; r < nr; r++) {
                zs_rank &q = R[(size_t)r];
                const int64_t c0 = q.lo ? std::max(q.e0, q.z0 - ZS_BLUR) : q.e0, c1 = q.hi ? std::min(q.e1, q.z1 + ZS_BLUR) : q.e1;
                const int64_t a = c0 - q.e0, b = c1 - q.e0, nzl = q.e1 - q.e0;
                ZS_HIP(hipSetDevice(q.dev));
                if (banded && (q.lo || q.hi)) {
                    if (q.lo) ZS_RC(blur_window_dev(q.c, q.L[j - 1], q.L[j], q.D[j - 1], X, Y, nzl, q.z0 - q.e0, q.z0 - q.e0 + hb, extras[j - 1], 0.01f));
                    if (q.hi) ZS_RC(blur_window_dev(q.c, q.L[j - 1], q.L[j], q.D[j - 1], X, Y, nzl, q.z1 - q.e0 - hb, q.z1 - q.e0, extras[j - 1], 0.01f));
                    ZS_HIP(hipEventRecord(q.ev_level, q.c->stream)); /* the bands are final: the neighbours may fetch them */
                } else {
                    /* level j on slab +- 8 (clipped to the buffer: at a face of the whole volume the buffer ends at the face,
                     * which is what makes the zero border exact), D_{j-1} fused */
                    ZS_RC(blur_dev(q.c, q.L[j - 1] + a * XY, q.L[j] + a * XY, q.D[j - 1] ? q.D[j - 1] + a * XY : nullptr, X, Y, b - a, extras[j - 1], 0.01f));
                    ZS_HIP(hipEventRecord(q.ev_level, q.c->stream));
                    if (j == 3) ZS_HIP(hipEventRecord(q.ev_l3, q.c->stream));
                }
            }
            if (banded)
                for (r = 0; r < nr; r++) { /* the interior, while the bands travel */
                    zs_rank &q = R[(size_t)r];
                    if (!q.lo && !q.hi) continue;
                    const int64_t w0 = q.lo ? q.z0 - q.e0 + hb : 0, w1 = q.hi ? q.z1 - q.e0 - hb : q.e1 - q.e0;
                    ZS_HIP(hipSetDevice(q.dev));
                    ZS_RC(blur_window_dev(q.c, q.L[j - 1], q.L[j], q.D[j - 1], X, Y, q.e1 - q.e0, w0, w1, extras[j - 1], 0.01f));
                    if (j == 3) ZS_HIP(hipEventRecord(q.ev_l3, q.c->stream));
                }
            /* the hb-slice halo of the new level from the two neighbours (their own slices, exact), queued behind the
             * sender's event -- on the receiver's halo stream beside its interior launch (bands first), or on its main
             * stream; then the fused DoG redone on the halo slices */
            ZS_X(zs_xfer_begin(h->tr));
            for (r = 0; r < nr; r++) {
                zs_rank &q = R[(size_t)r];
                const size_t bytes = sizeof(float) * (size_t)(hb * XY);
                hipStream_t hs = banded ? q.halo_stream : q.c->stream;
                if (q.lo) { /* global slices [z0 - hb, z0): the lower neighbour's last own slices */
                    zs_rank &p = R[(size_t)r - 1];
                    ZS_X(zs_xfer(h->tr, 0, r - 1, p.L[j] + (q.z0 - hb - p.e0) * XY, banded ? p.halo_stream : p.c->stream, p.ev_level, r,
                                 q.L[j] + (q.z0 - hb - q.e0) * XY, hs, (size_t)(hb * XY)));
                    st.halo_bytes_critical += (int64_t)bytes;
                    if (banded) st.halo_bytes_hidden += (int64_t)bytes;
                    st.exchanges++;
                }
                if (q.hi) { /* global slices [z1, z1 + hb): the upper neighbour's first own slices */
                    zs_rank &p = R[(size_t)r + 1];
                    ZS_X(zs_xfer(h->tr, 0, r + 1, p.L[j] + (q.z1 - p.e0) * XY, banded ? p.halo_stream : p.c->stream, p.ev_level, r,
                                 q.L[j] + (q.z1 - q.e0) * XY, hs, (size_t)(hb * XY)));
                    st.halo_bytes_critical += (int64_t)bytes;
                    if (banded) st.halo_bytes_hidden += (int64_t)bytes;
                    st.exchanges++;
                }
            }
            ZS_X(zs_xfer_end(h->tr)); /* RCCL queues the step's sends and receives here: what follows is behind them */
            for (r = 0; r < nr; r++) {
                zs_rank &q = R[(size_t)r];
                ZS_HIP(hipSetDevice(q.dev));
                if (banded && (q.lo || q.hi)) { /* the main stream goes on behind the arrivals */
                    ZS_COMM(hipEventRecord(q.ev_halo, q.halo_stream));
                    ZS_COMM(hipStreamWaitEvent(q.c->stream, q.ev_halo, 0));
                }
                const int64_t a = (q.lo ? std::max(q.e0, q.z0 - ZS_BLUR) : q.e0) - q.e0, b = (q.hi ? std::min(q.e1, q.z1 + ZS_BLUR) : q.e1) - q.e0;
                if (q.D[j - 1] && q.lo && q.z0 - q.e0 > a)
                    ZS_HIP(sift3d_launch_dog(q.c->stream, q.L[j - 1] + a * XY, q.L[j] + a * XY, q.D[j - 1] + a * XY, (q.z0 - q.e0 - a) * XY));
                if (q.D[j - 1] && q.hi && b > q.z1 - q.e0)
                    ZS_HIP(sift3d_launch_dog(q.c->stream, q.L[j - 1] + (q.z1 - q.e0) * XY, q.L[j] + (q.z1 - q.e0) * XY, q.D[j - 1] + (q.z1 - q.e0) * XY, (b - (q.z1 - q.e0)) * XY));
            }
            if (j == 3) {
                /* L1..L3 are final: the other 24 slices of their patch halos, on the copy stream, while L4, L5 and the
                 * extrema passes run on the main one */
                ZS_X(zs_xfer_begin(h->tr));
                for (r = 0; r < nr; r++) {
                    zs_rank &q = R[(size_t)r];
                    if (!q.lo && !q.hi) continue;
                    for (int l = 1; l <= 3; l++) {
                        if (q.lo) {
                            zs_rank &p = R[(size_t)r - 1];
                            const int64_t s0 = std::max(q.e0, q.z0 - ZS_HALO), s1 = q.z0 - ZS_BLUR;
                            if (s1 > s0) {
                                ZS_X(zs_xfer(h->tr, 1, r - 1, p.L[l] + (s0 - p.e0) * XY, p.copy_stream, p.ev_l3, r, q.L[l] + (s0 - q.e0) * XY, q.copy_stream,
                                             (size_t)((s1 - s0) * XY)));
                                st.halo_bytes_deferred += (int64_t)sizeof(float) * (s1 - s0) * XY;
                            }
                        }
                        if (q.hi) {
                            zs_rank &p = R[(size_t)r + 1];
                            const int64_t s0 = q.z1 + ZS_BLUR, s1 = std::min(q.e1, q.z1 + ZS_HALO);
                            if (s1 > s0) {
                                ZS_X(zs_xfer(h->tr, 1, r + 1, p.L[l] + (s0 - p.e0) * XY, p.copy_stream, p.ev_l3, r, q.L[l] + (s0 - q.e0) * XY, q.copy_stream,
                                             (size_t)((s1 - s0) * XY)));
                                st.halo_bytes_deferred += (int64_t)sizeof(float) * (s1 - s0) * XY;
                            }
                        }
                    }
                    st.exchanges++;
                }
                ZS_X(zs_xfer_end(h->tr));
                for (r = 0; r < nr; r++) {
                    zs_rank &q = R[(size_t)r];
                    if (!q.lo && !q.hi) continue;
                    ZS_HIP(hipSetDevice(q.dev));
                    ZS_HIP(hipEventRecord(q.ev_patch, q.copy_stream));
                }
            }
        }
        /* extrema of the rank's own slices; the level table in whole-volume terms */
        for (r = 0; r < nr; r++) {
            zs_rank &q = R[(size_t)r];
            ZS_HIP(hipSetDevice(q.dev));
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
                ZS_RC(cand_append(q.c, jb, true));
                sift3d_level &lv = q.levels[(size_t)id];
                lv.img = q.L[l + 1]; lv.dogc = q.D[l + 1];
                lv.X = (int)X; lv.Y = (int)Y; lv.Z = (int)zo; lv.XP = (int)X;
                lv.sigma_h = sig[l]; lv.sigma_c = sig[l + 1]; lv.sigma_l = sig[l + 2];
                lv.octave_factor = fscale;
                lv.Zl = (int)(q.e1 - q.e0); lv.z_off = (int)q.e0; lv.pad = 0;
            }
            if (q.lo || q.hi) ZS_HIP(hipStreamWaitEvent(q.c->stream, q.ev_patch, 0)); /* before the subsample reads L3 beyond +- 8 */
        }
        fscale *= 2.0f;
        if (o + 1 >= (int)plan.oct.size()) break;
        /* ---- level 0 of the next octave ---- */
        const int64_t Xn = plan.oct[(size_t)o + 1][0], Yn = plan.oct[(size_t)o + 1][1], zn = plan.oct[(size_t)o + 1][2], XYn = Xn * Yn;
        if (sharded && o + 1 < K) { /* next octave sharded too: slab +- 16 of L3 -> next slab +- 8 */
            for (r = 0; r < S; r++) {
                zs_rank &q = R[(size_t)r];
                int64_t n0, n1;
                plan.slab(r, o + 1, n0, n1);
                const int64_t ne0 = q.lo ? std::max<int64_t>(0, n0 - ZS_HALO) : n0, ne1 = q.hi ? std::min<int64_t>(zn, n1 + ZS_HALO) : n1;
                const int64_t s0 = q.lo ? std::max(ne0, n0 - ZS_BLUR) : ne0, s1 = q.hi ? std::min(ne1, n1 + ZS_BLUR) : ne1;
                ZS_HIP(hipSetDevice(q.dev));
                float *nx0 = q.alloc((ne1 - ne0) * XYn);
                if (!nx0) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "rank %d: out of device memory", r); goto done; }
                ZS_HIP(sift3d_launch_subsample(q.c->stream, q.L[3] + (2 * s0 - q.e0) * XY, X, X, Y, 2 * (s1 - s0), nx0 + (s0 - ne0) * XYn, Xn));
                next0[r] = nx0;
            }
        } else if (sharded) { /* last sharded octave: every rank subsamples exactly its slab, rank 0 assembles the whole octave */
            zs_rank &root = R[0];
            ZS_HIP(hipSetDevice(root.dev));
            float *full = root.alloc(zn * XYn);
            if (!full) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "rank 0: out of device memory"); goto done; }
            int64_t at = 0;
            struct gather_part { int rank; const float *src; int64_t at, t; };
            std::vector<gather_part> parts;
            for (r = 0; r < S; r++) {
                zs_rank &q = R[(size_t)r];
                const int64_t t = std::min<int64_t>((q.z1 - q.z0) / 2, zn - at); /* an odd last slice of the whole volume is dropped, as in the serial code */
                if (t <= 0) continue;
                ZS_HIP(hipSetDevice(q.dev));
                float *part = r == 0 ? full + at * XYn : q.alloc(t * XYn);
                if (!part) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "rank %d: out of device memory", r); goto done; }
                ZS_HIP(sift3d_launch_subsample(q.c->stream, q.L[3] + (q.z0 - q.e0) * XY, X, X, Y, 2 * t, part, Xn));
                if (r > 0) {
                    ZS_HIP(hipEventRecord(q.ev_level, q.c->stream));
                    parts.push_back({r, part, at, t});
                }
                at += t;
            }
            ZS_X(zs_xfer_begin(h->tr)); /* every rank's part of the octave to rank 0, ordered in rank 0's main stream */
            for (const gather_part &g : parts) {
                zs_rank &q = R[(size_t)g.rank];
                ZS_X(zs_xfer(h->tr, 0, g.rank, g.src, q.c->stream, q.ev_level, 0, full + g.at * XYn, root.c->stream, (size_t)(g.t * XYn)));
                st.gather_bytes += (int64_t)sizeof(float) * g.t * XYn;
            }
            ZS_X(zs_xfer_end(h->tr));
            if (at != zn) { rc = SIFT3D_ERR_ARG; snprintf(errbuf, sizeof errbuf, "slab plan does not tile octave %d (%lld of %lld slices)", o + 1, (long long)at, (long long)zn); goto done; }
            next0[0] = full;
        } else { /* unsharded: rank 0 alone */
            zs_rank &q = R[0];
            ZS_HIP(hipSetDevice(q.dev));
            float *nx0 = q.alloc(zn * XYn);
            if (!nx0) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "rank 0: out of device memory"); goto done; }
            ZS_HIP(sift3d_launch_subsample(q.c->stream, q.L[3], X, X, Y, zo, nx0, Xn));
            next0[0] = nx0;
        }
    }

    /* ---- per-keypoint stage on every rank, phase by phase across the ranks so that no device waits for another's host
     * round trip: (1) every rank's extrema count is requested; (2) rank by rank the count is awaited, the candidates are
     * sorted and the keypoint side of the stage is queued -- the devices before it are already computing; (3) rank by rank
     * the descriptor launches follow the keypoint chunks; (4) one synchronisation per rank at the end. ---- */
    {
        std::vector<int64_t> ncands((size_t)S, 0);
        for (r = 0; r < S; r++) {
            zs_rank &q = R[(size_t)r];
            ZS_HIP(hipSetDevice(q.dev));
            /* (the main stream already waits for the deferred patch halos of every sharded octave: ev_patch above) */
            ZS_RC(cand_count_queue(q.c));
        }
        for (r = 0; r < S; r++) {
            zs_rank &q = R[(size_t)r];
            ZS_HIP(hipSetDevice(q.dev));
            ZS_RC(cand_finalize(q.c, &ncands[(size_t)r]));
            st.n_extrema += ncands[(size_t)r];
            ZS_RC(describe_queue(q.c, q.levels, ncands[(size_t)r], desc_mode, eig_thres, size_factor, false));
        }
        for (r = 0; r < S; r++) {
            ZS_HIP(hipSetDevice(R[(size_t)r].dev));
            ZS_RC(describe_launch(R[(size_t)r].c));
        }
        for (r = 0; r < S; r++) {
            zs_rank &q = R[(size_t)r];
            int64_t nrec = 0;
            ZS_HIP(hipSetDevice(q.dev));
            ZS_RC(describe_finish(q.c, &nrec));
            recs[(size_t)r].assign(q.c->h_recs, q.c->h_recs + nrec);
            grps[(size_t)r].assign(q.c->h_group, q.c->h_group + nrec);
            st.n_keypoints += q.c->last.n_keypoints;
        }
    }
    {
        /* merge: within a group (level, is_max) slabs are in z order, so rank order is the serial raster order */
        int64_t total = 0;
        for (r = 0; r < S; r++) total += (int64_t)recs[(size_t)r].size();
        sift3d_feature *res = (sift3d_feature *)malloc(sizeof(sift3d_feature) * (size_t)(total ? total : 1));
        if (!res) { rc = SIFT3D_ERR_MEMORY; snprintf(errbuf, sizeof errbuf, "out of host memory"); goto done; }
        std::vector<int64_t> count(2 * 96 + 2, 0);
        for (r = 0; r < S; r++)
            for (int g : grps[(size_t)r]) count[(size_t)(g < 0 || g >= 192 ? 192 : g) + 1]++;
        for (size_t g = 1; g < count.size(); g++) count[g] += count[g - 1];
        for (r = 0; r < S; r++)
            for (size_t i = 0; i < recs[(size_t)r].size(); i++) {
                const int g = grps[(size_t)r][i];
                res[count[(size_t)(g < 0 || g >= 192 ? 192 : g)]++] = recs[(size_t)r][i];
            }
        *out = res;
        *n_out = total;
        st.n_records = total;
    }

done:
    for (size_t i = 0; i < R.size(); i++) { /* everything queued has to be done before the buffers go back */
        zs_rank &q = R[i];
        hipSetDevice(q.dev);
        hipStreamSynchronize(q.c->stream);
        hipStreamSynchronize(q.copy_stream);
        hipStreamSynchronize(q.halo_stream);
    }
    for (size_t i = 0; i < R.size(); i++) {
        hipSetDevice(R[i].dev);
        R[i].recycle();
    }
    st.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (stats) *stats = st;
    if (rc != SIFT3D_OK && err && err_len > 0) snprintf(err, (size_t)err_len, "%s", errbuf);
    return rc;
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
