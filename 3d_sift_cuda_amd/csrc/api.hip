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
 * blocks of the C-ABI.  The one-process Z-slab driver (sift3d_zslab_*) is
 * zslab_driver.hip; what the two share is pipeline.h.
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

#include "pipeline.h"

int set_err(sift3d_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

/* entry points that work in the context's own level buffers: not on a slab context, which has none */
#define NEED_LEVELS(c)                                                                                                  \
    do {                                                                                                                \
        if ((c) && (c)->lean) return set_err((c), SIFT3D_ERR_ARG, "%s needs a full context (sift3d_create), not a slab context", __func__); \
    } while (0)

extern "C" int sift3d_abi_version(void) { return SIFT3D_ABI_VERSION; }

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
    hipFree(c->place.d_counts);
    hipFree(c->place.d_shift);
    if (c->place.h_counts) hipHostFree(c->place.h_counts);
}

/* octave list of a volume: halve while every dimension stays above 2 (MultiScale.cpp:359-360,546-556) */

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

/* (Re)size the candidate, sort and key buffers.  The new buffers are made first and the old ones released only when every
 * allocation succeeded: a failed growth leaves the context as it was (old buffers, old capacity) and reports SIFT3D_ERR_MEMORY,
 * so a caller that ignores the result of sift3d_reserve still runs on valid buffers (round-5 advisor finding). */
static int alloc_cands(sift3d_ctx *c, int64_t cap)
{
    unsigned long long *ka = nullptr, *kb = nullptr;
    sift3d_cval *va = nullptr, *vb = nullptr;
    void *tmp = nullptr;
    const size_t tmp_bytes = sift3d_sort_temp_bytes(cap) + 256;
    const bool ok = hipMalloc((void **)&ka, sizeof(unsigned long long) * (size_t)cap) == hipSuccess &&
                    hipMalloc((void **)&kb, sizeof(unsigned long long) * (size_t)cap) == hipSuccess &&
                    hipMalloc((void **)&va, sizeof(sift3d_cval) * (size_t)cap) == hipSuccess &&
                    hipMalloc((void **)&vb, sizeof(sift3d_cval) * (size_t)cap) == hipSuccess && hipMalloc(&tmp, tmp_bytes) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(tmp);
        return SIFT3D_ERR_MEMORY;
    }
    hipFree(c->keys_a); hipFree(c->keys_b); hipFree(c->vals_a); hipFree(c->vals_b); hipFree(c->sort_tmp);
    c->keys_a = ka; c->keys_b = kb;
    c->vals_a = va; c->vals_b = vb;
    c->sort_tmp = tmp;
    c->sort_tmp_bytes = tmp_bytes;
    c->cand_cap = cap;
    return SIFT3D_OK;
}

static void destroy_sync_objects(sift3d_ctx *c)
{
    hipStream_t streams[] = {c->ex_stream, c->ex_stream2, c->kp_stream};
    for (hipStream_t st : streams)
        if (st) {
            hipStreamSynchronize(st);
            hipStreamDestroy(st);
        }
    hipEvent_t events[] = {c->ev_ex2[0], c->ev_ex2[1], c->ev_reset, c->ev_desc, c->ev_oct[0], c->ev_oct[1], c->ev_fence[0], c->ev_fence[1],
                           c->ev_split[0], c->ev_split[1], c->ev_split[2]};
    for (hipEvent_t e : events)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_kpc)
        if (e) hipEventDestroy(e);
    if (c->h_cnt0) hipHostFree(c->h_cnt0);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
}

/* lean: a slab context (sift3d_create_slab) -- the caller owns the level buffers; only the pass intermediates, the
 * candidate lists and the per-keypoint buffers live here */
sift3d_ctx *ctx_create(int device, int64_t nx, int64_t ny, int64_t nz, bool lean)
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
    c->tune[SIFT3D_TUNE_FUSED_SUB] = 1;
    c->tune[SIFT3D_TUNE_SPLIT_TAIL] = 1;
    c->tune[SIFT3D_TUNE_DESC_SEGMENT] = 32;
    /* every octave of a capN volume, back to back: capN * (1 + 1/8 + 1/64 + ...) plus alignment */
    c->capTot = c->capN + c->capN / 7 + 4 * ny * nz + 64 * 34; /* + up to three pad columns per row of every coarser octave */
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    hipStream_t *streams[] = {&c->ex_stream, &c->ex_stream2, &c->kp_stream};
    for (hipStream_t *st : streams) ok = ok && hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t *events[] = {&c->ev_ex2[0], &c->ev_ex2[1], &c->ev_reset, &c->ev_desc, &c->ev_oct[0], &c->ev_oct[1], &c->ev_fence[0], &c->ev_fence[1],
                            &c->ev_split[0], &c->ev_split[1], &c->ev_split[2]};
    for (hipEvent_t *e : events) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    for (hipEvent_t &e : c->ev_kpc) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_cnt0, sizeof(unsigned long long) * (16 + SIFT3D_KP_MAX_CHUNKS), hipHostMallocDefault) == hipSuccess;
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
    ok = ok && hipMalloc((void **)&c->d_count, sizeof(unsigned long long) * 8) == hipSuccess;
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
    static const int lo[SIFT3D_TUNE_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0},
                     hi[SIFT3D_TUNE_COUNT] = {2, 4096, 2, 1, 1, 64, SIFT3D_KP_MAX_CHUNKS, 1, 1 + SIFT3D_MAX_FRAMES, 2, 1, 2, 1 << 20, 3, 2};
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
    void add_bytes(double more) /* the launch turned out to do more (the subsample riding on the level-3 blur) */
    {
        c->last.alg_bytes[stage] += more;
        bytes += more;
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

void timing_begin(sift3d_ctx *c)
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
/* out = blur(in); if dog != NULL also dog = in - out.  out may be NULL when only the DoG is wanted.  Uses T[0], T[1].
 * sub (optional): the next octave's level 0, the 2 x 2 x 2 mean of out as a dense (X / 2) x (Y / 2) x (Z / 2) volume; written
 * only where the fused launch can carry it, *sub_done says whether -- the caller launches the subsample itself if not. */
int blur_dev(sift3d_ctx *c, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, float sigma,
                    float min_value, float *sub, bool *sub_done)
{
    if (sub_done) *sub_done = false;
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
    const sift3d_blur_tuning bt = {c->tune[SIFT3D_TUNE_FUSED_CHUNKS], c->tune[SIFT3D_TUNE_FUSED_ROWS], c->tune[SIFT3D_TUNE_FUSED_TILE], c->tune[SIFT3D_TUNE_FUSED_ORDER], c->tune[SIFT3D_TUNE_FUSED_STAGGER]};
    /* measured standalone (tools/bench_blur_ab.sh 128 / 64): below 2^22 voxels the one launch still beats the three for 7 and
     * 9 taps (0.020 / 0.026 against 0.042 / 0.043 ms at 128^3), ties at 11-13 and loses at 17 */
    if (fmode == 2 || (fmode == 1 && (N >= (double)(1 << 22) || (N >= (double)(1 << 18) && n <= 9)))) {
        stage_scope sc(c, SIFT3D_STAGE_BLUR_FUSED, (dog && out ? 12.0 : 8.0) * N, n, (int64_t)N, ws);
        int with_sub = 0;
        hipError_t e = sift3d_launch_blur_fused(ws, in, out, dog, X, Y, Z, taps, n, &bt, 0, -1, c->tune[SIFT3D_TUNE_FUSED_SUB] ? sub : nullptr, &with_sub);
        if (e == hipSuccess) {
            if (with_sub) {
                sc.add_bytes(0.5 * N); /* one float stored per eight voxels */
                if (sub_done) *sub_done = true;
            }
            return SIFT3D_OK;
        }
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
bool blur_window_supported(int64_t X, int64_t Y, float sigma, float min_value)
{
    float taps[SIFT3D_MAX_TAPS];
    const int n = sift3d_gauss_taps(sigma, min_value, taps);
    return n >= 3 && n <= 2 * SIFT3D_FAST_MAX_R + 1 && X % 4 == 0 && X * Y < (1ll << 29);
}

int blur_window_dev(sift3d_ctx *c, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1,
                           float sigma, float min_value)
{
    float taps[SIFT3D_MAX_TAPS];
    const int n = sift3d_gauss_taps(sigma, min_value, taps);
    if (n < 3 || zo0 < 0 || zo1 > Z || zo1 <= zo0) return set_err(c, SIFT3D_ERR_ARG, "bad blur window [%lld, %lld) of %lld planes", (long long)zo0, (long long)zo1, (long long)Z);
    const sift3d_blur_tuning bt = {c->tune[SIFT3D_TUNE_FUSED_CHUNKS], c->tune[SIFT3D_TUNE_FUSED_ROWS], c->tune[SIFT3D_TUNE_FUSED_TILE], c->tune[SIFT3D_TUNE_FUSED_ORDER], c->tune[SIFT3D_TUNE_FUSED_STAGGER]};
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

/* level + DoG + the half-size volume the next octave starts from, as the pyramid produces them at level 3 */
extern "C" int sift3d_gauss_blur_dog_half_dev(sift3d_ctx *c, const float *d_in, float *d_out, float *d_dog, float *d_half, int64_t nx,
                                              int64_t ny, int64_t nz, float sigma, float min_value, int *in_one_launch)
{
    if (in_one_launch) *in_one_launch = 0;
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!d_in || !d_out || !d_half) return set_err(c, SIFT3D_ERR_ARG, "null pointer");
    if (nx < 2 || ny < 2 || nz < 2) return set_err(c, SIFT3D_ERR_ARG, "subsample needs every dimension >= 2");
    HIPCHK(c, hipSetDevice(c->device));
    rc = fence_in(c);
    if (rc) return rc;
    bool carried = false;
    /* the half-size volume is dense here (rows of nx / 2): the launch can carry it when those rows are whole 16-byte vectors */
    rc = blur_dev(c, d_in, d_out, d_dog, nx, ny, nz, sigma, min_value, nx % 8 == 0 ? d_half : nullptr, &carried);
    if (rc) return rc;
    if (!carried) {
        stage_scope sc(c, SIFT3D_STAGE_SUBSAMPLE, 4.5 * (double)nx * ny * nz, 0, nx * ny * nz);
        HIPCHK(c, sift3d_launch_subsample(c->stream, d_out, nx, nx, ny, nz, d_half, nx / 2));
    }
    if (in_one_launch) *in_one_launch = carried ? 1 : 0;
    return fence_out(c);
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
int cand_reset(sift3d_ctx *c, hipStream_t on)
{
    if (!on) on = c->stream;
    c->jobs.clear();
    c->cand_split_at = 0;
    c->cand_group = 0;
    HIPCHK(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned long long) * 8, on));
    /* every extrema pass of the run gets its own counter set: one memset here instead of one per pass */
    HIPCHK(c, hipMemsetAsync(c->surv_counts, 0, sizeof(unsigned long long) * SIFT3D_SURV_COUNTERS * SIFT3D_SURV_SETS, on));
    HIPCHK(c, hipMemsetAsync(c->list2_counts, 0, sizeof(unsigned long long) * SIFT3D_LIST2_COUNTERS * SIFT3D_SURV_SETS, on));
    HIPCHK(c, hipMemsetAsync(c->d_rec_base, 0, sizeof(int), on)); /* first record of the per-keypoint stage's first chunk (the split tail
                                                                    * does not clear it again when the first count arrives) */
    c->surv_set = 0;
    return SIFT3D_OK;
}

/* where the extrema launches queued now append: the whole list, or the part of it the current group owns (split tail) */
struct cand_target {
    unsigned long long *keys;
    sift3d_cval *vals;
    unsigned long long *count;
    int64_t cap;
};
static cand_target cand_target_of(const sift3d_ctx *c)
{
    if (c->cand_split_at <= 0) return {c->keys_a, c->vals_a, c->d_count, c->cand_cap};
    if (c->cand_group == 0) return {c->keys_a, c->vals_a, c->d_count, c->cand_split_at};
    return {c->keys_a + c->cand_split_at, c->vals_a + c->cand_split_at, c->d_count + 4, c->cand_cap - c->cand_split_at};
}

int cand_append(sift3d_ctx *c, const level_job &j, bool record)
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
    const cand_target tg = cand_target_of(c);
    HIPCHK(c, sift3d_launch_extrema(st, j.dp, j.dc, j.dn, j.X, j.Xl ? j.Xl : j.X, j.Y, j.Z, j.z_lo, j.z_hi, j.lvl_id, tg.keys,
                                    tg.vals, tg.count, tg.cap, surv, counters, c->d_count + 2, cover, !fresh,
                                    lazy ? &lz : nullptr));
    return SIFT3D_OK;
}

static int cand_replay(sift3d_ctx *c)
{
    c->cand_split_at = 0; /* a replay fills one list, whatever the first attempt did */
    c->cand_group = 0;
    HIPCHK(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned long long) * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->surv_counts, 0, sizeof(unsigned long long) * SIFT3D_SURV_COUNTERS * SIFT3D_SURV_SETS, c->stream));
    HIPCHK(c, hipMemsetAsync(c->list2_counts, 0, sizeof(unsigned long long) * SIFT3D_LIST2_COUNTERS * SIFT3D_SURV_SETS, c->stream));
    c->surv_set = 0;
    for (const level_job &j : c->jobs) {
        int rc = cand_append(c, j, false);
        if (rc) return rc;
    }
    return SIFT3D_OK;
}

/* An own-level list was cut short (high_water = the length it would have needed): from now on the lists of this context are
 * sized for the worst case of a level, and the first one is grown to the mark.  Nothing may be running on the context. */
static int surv_make_room(sift3d_ctx *c, unsigned long long high_water)
{
    c->surv_div = 1;
    if ((int64_t)high_water > c->surv_cap) {
        hipFree(c->surv);
        c->surv = nullptr;
        c->surv_cap = (int64_t)high_water + (int64_t)high_water / 4 + 4096;
        HIPCHK(c, hipMalloc((void **)&c->surv, sizeof(sift3d_survivor) * (size_t)c->surv_cap));
    }
    return SIFT3D_OK;
}

/* The count of validated extrema comes back in two steps so that a driver with several contexts can queue the read-back
 * on all of them before it waits for the first: cand_count_queue (asynchronous), cand_finalize (waits, replays the extrema
 * launches into bigger lists if one overflowed, sorts). */
int cand_count_queue(sift3d_ctx *c)
{
    unsigned long long *cnt = c->h_cnt0 + 4; /* validated extrema, survivors of the last level, survivor overflow */
    cnt[0] = cnt[1] = cnt[2] = 0;
    HIPCHK(c, hipMemcpyAsync(cnt, c->d_count, sizeof(unsigned long long) * 3, hipMemcpyDeviceToHost, c->stream));
    c->count_queued = true;
    return SIFT3D_OK;
}

int cand_finalize(sift3d_ctx *c, int64_t *count_out)
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
            int rc = surv_make_room(c, cnt[2]);
            if (rc) return rc;
            rc = cand_replay(c);
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
/* before a volume of this shape is copied into c->vol: the pad columns of pitched rows cleared (once per geometry) */
static int load_volume_prepare(sift3d_ctx *c, int64_t nx, int64_t ny, int64_t nz)
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
    if (xp == nx) c->pad_nx = 0; /* dense rows overwrite what would be pad columns of another geometry */
    return SIFT3D_OK;
}

static int load_volume(sift3d_ctx *c, const float *src, bool from_host, int64_t nx, int64_t ny, int64_t nz)
{
    const int64_t xp = pitch_of(nx);
    const int rcp = load_volume_prepare(c, nx, ny, nz);
    if (rcp) return rcp;
    if (xp == nx) {
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

/* The volume in runs of whole z planes (round 5: featExtract uploads what it has read while the rest of the file is still
 * being read or inflated).  begin: the shape that will arrive (and resize as in sift3d_set_volume_resized); planes: planes
 * [z0, z0 + n) from host memory, queued on the context's stream -- any order, every plane exactly once; end: the resize
 * launch if one was asked for, then waits until the volume is resident.  Equivalent to sift3d_set_volume[_resized] of the
 * assembled volume. */
extern "C" int sift3d_set_volume_begin(sift3d_ctx *c, int64_t nx, int64_t ny, int64_t nz, int resize)
{
    NEED_LEVELS(c);
    if (!c || nx < 1 || ny < 1 || nz < 1 || nx * ny * nz > c->capN) return set_err(c, SIFT3D_ERR_ARG, "set_volume_begin: bad shape");
    if (resize != 0 && (nx < 2 || ny < 2 || nz < 2)) return set_err(c, SIFT3D_ERR_ARG, "set_volume_begin: bad shape for a resize");
    const int64_t ox = resize > 0 ? 2 * nx : (resize < 0 ? nx / 2 : nx), oy = resize > 0 ? 2 * ny : (resize < 0 ? ny / 2 : ny),
                  oz = resize > 0 ? 2 * nz : (resize < 0 ? nz / 2 : nz);
    int rc = check_shape(c, ox, oy, oz);
    if (rc) return rc;
    if (oz <= 1) return set_err(c, SIFT3D_ERR_ARG, "Could not read volume (z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    c->has_volume = false;
    if (resize != 0) {
        rc = ensure_T(c, ox * oy * oz > nx * ny * nz ? ox * oy * oz : nx * ny * nz);
        if (rc) return rc;
    } else {
        rc = load_volume_prepare(c, nx, ny, nz);
        if (rc) return rc;
    }
    c->up.open = true;
    c->up.nx = nx; c->up.ny = ny; c->up.nz = nz;
    c->up.got = 0;
    c->up.resize = resize;
    c->up.seen.assign((size_t)nz, false);
    return SIFT3D_OK;
}

extern "C" int sift3d_set_volume_planes(sift3d_ctx *c, const float *planes, int64_t z0, int64_t n)
{
    if (!c || !c->up.open) return set_err(c, SIFT3D_ERR_ARG, "set_volume_planes without set_volume_begin");
    if (!planes || z0 < 0 || n < 1 || z0 + n > c->up.nz) {
        c->up.open = false; /* a caller that hands over planes the volume does not have starts again */
        return set_err(c, SIFT3D_ERR_ARG, "set_volume_planes: planes [%lld, %lld) of %lld", (long long)z0, (long long)(z0 + n), (long long)c->up.nz);
    }
    for (int64_t z = z0; z < z0 + n; z++)
        if (c->up.seen[(size_t)z]) { /* counting planes alone would accept a plane twice in place of one that never came */
            c->up.open = false;
            return set_err(c, SIFT3D_ERR_ARG, "set_volume_planes: plane %lld arrived twice", (long long)z);
        }
    for (int64_t z = z0; z < z0 + n; z++) c->up.seen[(size_t)z] = true;
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t nx = c->up.nx, ny = c->up.ny, xp = pitch_of(nx);
    if (c->up.resize != 0) {
        HIPCHK(c, hipMemcpyAsync(c->T[0] + z0 * nx * ny, planes, sizeof(float) * (size_t)(n * nx * ny), hipMemcpyHostToDevice, c->stream));
    } else if (xp == nx) {
        HIPCHK(c, hipMemcpyAsync(c->vol + z0 * nx * ny, planes, sizeof(float) * (size_t)(n * nx * ny), hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(c, hipMemcpy2DAsync(c->vol + z0 * xp * ny, sizeof(float) * (size_t)xp, planes, sizeof(float) * (size_t)nx, sizeof(float) * (size_t)nx,
                                   (size_t)(ny * n), hipMemcpyHostToDevice, c->stream));
    }
    c->up.got += n;
    return SIFT3D_OK;
}

extern "C" int sift3d_set_volume_end(sift3d_ctx *c)
{
    if (!c || !c->up.open) return set_err(c, SIFT3D_ERR_ARG, "set_volume_end without set_volume_begin");
    c->up.open = false;
    if (c->up.got != c->up.nz) return set_err(c, SIFT3D_ERR_ARG, "set_volume_end: %lld of %lld planes arrived", (long long)c->up.got, (long long)c->up.nz);
    HIPCHK(c, hipSetDevice(c->device));
    int64_t nx = c->up.nx, ny = c->up.ny, nz = c->up.nz;
    if (c->up.resize != 0) {
        const int64_t ox = c->up.resize > 0 ? 2 * nx : nx / 2, oy = c->up.resize > 0 ? 2 * ny : ny / 2, oz = c->up.resize > 0 ? 2 * nz : nz / 2;
        if (c->up.resize > 0) HIPCHK(c, sift3d_launch_double_size(c->stream, c->T[0], nx, ny, nz, c->T[1]));
        else HIPCHK(c, sift3d_launch_halve_size(c->stream, c->T[0], nx, ny, nz, c->T[1]));
        const int rc = load_volume(c, c->T[1], false, ox, oy, oz);
        if (rc) return rc;
        nx = ox; ny = oy; nz = oz;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* the caller may free its planes */
    c->nx = nx; c->ny = ny; c->nz = nz;
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
static int ensure_kp_buffers(sift3d_ctx *c, int64_t ncand, int64_t host_for = -1) /* host_for: candidates the pinned buffers are sized for, if fewer */
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
    if (host_for < 0 || host_for > ncand) host_for = ncand;
    if (c->hrecs_cap < host_for * per) { /* nothing of an earlier run is in flight here: describe_finish has synchronised */
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->kp_stream));
        int rc = ensure_host_records(c, host_for * per, 0);
        if (rc) return rc;
    }
    return SIFT3D_OK;
}

/* The buffers a run with about n_extrema validated extrema needs beyond the context's own -- keypoint records, identity
 * patches, record map, the pinned download buffers -- made now rather than inside the first extraction, where they cost a
 * process that extracts once (the command line) some 25 ms at 512^3.  A run that finds more grows them as before. */
extern "C" int sift3d_reserve(sift3d_ctx *c, int64_t n_extrema)
{
    if (!c || n_extrema < 0) return set_err(c, SIFT3D_ERR_ARG, "sift3d_reserve: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (n_extrema > c->cand_cap && alloc_cands(c, n_extrema + n_extrema / 4 + 4096) != SIFT3D_OK)
        return set_err(c, SIFT3D_ERR_MEMORY, "candidate buffer could not be grown to %lld entries", (long long)n_extrema);
    return ensure_kp_buffers(c, n_extrema, n_extrema);
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
    p.desc_seg = c->tune[SIFT3D_TUNE_DESC_SEGMENT] * 8;
    p.rec_shift = nullptr;
}

static int kp_chunks_for(const sift3d_ctx *c, int64_t ncand)
{
    int n = c->tune[SIFT3D_TUNE_KP_CHUNKS];
    if (n <= 0) n = SIFT3D_KP_DEFAULT_CHUNKS;
    if (n > SIFT3D_KP_MAX_CHUNKS) n = SIFT3D_KP_MAX_CHUNKS;
    if (n > ncand) n = ncand > 0 ? (int)ncand : 1;
    return n;
}

/* the stage's state cleared and its two patch filters checked; taps3: the keypoint kernel's 3-tap filter */
static int describe_begin(sift3d_ctx *c, size_t nlevels, float *taps3)
{
    if (sift3d_gauss_taps(0.5f, 0.01f, taps3) != 3 || sift3d_gauss_taps((float)0.95, (float)0.01, c->kp.taps5) != 5)
        return set_err(c, SIFT3D_ERR_ARG, "unexpected patch tap counts");
    if (nlevels > 96) return set_err(c, SIFT3D_ERR_ARG, "too many levels");
    c->kp.ncand = 0;
    c->kp.nchunks = 0;
    c->kp.launched = 0;
    c->kp.nrec = 0;
    c->kp.split = false;
    c->place.dst = nullptr; /* a shared destination holds for one run */
    return SIFT3D_OK;
}

void describe_want_group_counts(sift3d_ctx *c, bool on) { c->place.counts = on; }

static int ensure_place_buffers(sift3d_ctx *c)
{
    if (c->place.d_counts) return SIFT3D_OK;
    HIPCHK(c, hipMalloc((void **)&c->place.d_counts, sizeof(int) * SIFT3D_GROUPS));
    HIPCHK(c, hipMalloc((void **)&c->place.d_shift, sizeof(int) * SIFT3D_GROUPS));
    HIPCHK(c, hipHostMalloc((void **)&c->place.h_counts, sizeof(int) * SIFT3D_GROUPS, hipHostMallocDefault));
    return SIFT3D_OK;
}

int describe_group_counts(sift3d_ctx *c, const int **counts, int64_t *total)
{
    if (!c->place.counts) return set_err(c, SIFT3D_ERR_ARG, "describe_group_counts: not asked for before describe_queue");
    {
        const int rc = ensure_place_buffers(c); /* (a run without candidates has not made them) */
        if (rc) return rc;
    }
    *total = 0;
    if (c->kp.ncand <= 0 || c->kp.nchunks != 1) { /* nothing queued (no candidates): every count is zero */
        memset(c->place.h_counts, 0, sizeof(int) * SIFT3D_GROUPS);
        *counts = c->place.h_counts;
        return c->kp.nchunks > 1 ? set_err(c, SIFT3D_ERR_ARG, "describe_group_counts: the stage runs in chunks") : SIFT3D_OK;
    }
    HIPCHK(c, hipEventSynchronize(c->ev_kpc[0]));
    for (int g = 0; g < SIFT3D_GROUPS; g++) *total += c->place.h_counts[g];
    *counts = c->place.h_counts;
    return SIFT3D_OK;
}

int describe_placement(sift3d_ctx *c, sift3d_feature *shared_list, const int *shift)
{
    if (!shared_list || !shift || !c->place.d_shift) return set_err(c, SIFT3D_ERR_ARG, "describe_placement: bad arguments");
    /* (pageable source: the copy has been staged when the call returns, the caller's table may go) */
    HIPCHK(c, hipMemcpyAsync(c->place.d_shift, shift, sizeof(int) * SIFT3D_GROUPS, hipMemcpyHostToDevice, c->stream));
    c->place.dst = shared_list;
    return SIFT3D_OK;
}

/* chunk i of the stage = candidates [a, b) of the sorted list, on stream ks: keypoint kernel, scan of the record counts, record
 * map (which leaves the chunk's first-record index for the next chunk on the device), read-back of the chunk's end index */
static int describe_queue_chunk(sift3d_ctx *c, int i, int64_t a, int64_t b, hipStream_t ks, const float *taps3)
{
    int *h_end = reinterpret_cast<int *>(c->h_cnt0 + 8); /* pinned: first record past chunk i */
    c->kp.first[i] = a;
    c->kp.first[i + 1] = b;
    h_end[i] = 0;
    {
        sift3d_kp_params q = c->kp.p;
        q.patch0 = c->patch0 + (size_t)a * SIFT3D_PATCH_VOX; /* the kernel indexes everything by its block number */
        stage_scope sc(c, SIFT3D_STAGE_KEYPOINT, 0.0, 0, b - a, ks);
        HIPCHK(c, sift3d_launch_keypointsA(ks, q, c->keys_b + a, c->vals_b + a, b - a, c->kps + a, c->nrec + a, taps3));
    }
    HIPCHK(c, sift3d_scan_counts(ks, c->scan_tmp, c->scan_tmp_bytes, c->nrec + a, c->offs + a, b - a));
    HIPCHK(c, sift3d_launch_recmap(ks, c->nrec + a, c->offs + a, b - a, (int)a, c->d_rec_base + i, c->rec_kp, c->rec_frame, c->d_count + 3));
    HIPCHK(c, hipMemcpyAsync(&h_end[i], c->d_rec_base + i + 1, sizeof(int), hipMemcpyDeviceToHost, ks));
    if (c->place.counts && i == 0) { /* records per group, for a driver that places several contexts' records in one list */
        int rc = ensure_place_buffers(c);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->place.d_counts, 0, sizeof(int) * SIFT3D_GROUPS, ks));
        HIPCHK(c, sift3d_launch_group_counts(ks, c->keys_b + a, c->nrec + a, b - a, c->place.d_counts));
        HIPCHK(c, hipMemcpyAsync(c->place.h_counts, c->place.d_counts, sizeof(int) * SIFT3D_GROUPS, hipMemcpyDeviceToHost, ks));
    }
    HIPCHK(c, hipEventRecord(c->ev_kpc[i], ks));
    return SIFT3D_OK;
}

int describe_queue(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand, int desc_mode, float eig_thres,
                          float size_factor, bool levels_on_device)
{
    float taps3[SIFT3D_MAX_TAPS];
    int rc0 = describe_begin(c, levels.size(), taps3);
    if (rc0) return rc0;
    c->kp.ncand = ncand;
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
    for (int i = 0; i < nch; i++) {
        rc = describe_queue_chunk(c, i, ncand * i / nch, ncand * (i + 1) / nch, c->stream, taps3);
        if (rc) return rc;
    }
    return SIFT3D_OK;
}

int describe_launch(sift3d_ctx *c)
{
    const int nch = c->kp.nchunks;
    if (nch <= 0) return SIFT3D_OK;
    const int *h_end = reinterpret_cast<const int *>(c->h_cnt0 + 8);
    /* one chunk: the descriptor kernel follows on the main stream; several: on the second stream, beside the next chunk's
     * keypoint kernel.  With every launch bracketed by events (timing modes 1 and 3) the launches stay on the main stream,
     * so that an event pair times its kernel alone. */
    /* split tail: the chunks were queued on kp_stream, the second one (the coarse octaves' few hundred extrema: a keypoint launch
     * of pure latency, 0.25 ms at 512^3) behind the pyramid's last launch.  The first chunk's descriptor launch goes to the main
     * stream, idle by then, and the second chunk's keypoint kernel and descriptors run beside it on kp_stream. */
    const bool split = c->kp.split && nch > 1;
    const bool beside = split || (!c->kp.split && nch > 1 && !(c->timing == 1 || c->timing == 3));
    int64_t base = 0;
    for (int i = 0; i < nch; i++) {
        hipStream_t ds = (split ? i > 0 : beside) ? c->kp_stream : c->stream;
        HIPCHK(c, hipEventSynchronize(c->ev_kpc[i]));
        const int64_t end = h_end[i], m = end - base;
        if (end < base || end > c->recs_cap) return set_err(c, SIFT3D_ERR_DEVICE, "record map out of range (%lld of %lld)", (long long)end, (long long)c->recs_cap);
        if (end > c->hrecs_cap && !c->place.dst) {
            /* more records than the pinned buffers were sized for: wait for the descriptor launches of the earlier chunks
             * (they store into the buffers about to be replaced), grow, carry their records over */
            HIPCHK(c, hipStreamSynchronize(ds));
            if (split) HIPCHK(c, hipStreamSynchronize(c->stream)); /* the first chunk's launch is on the main stream */
            /* chunks still to come: assume they yield records at the rate seen so far */
            const int64_t done_cand = c->kp.first[i + 1], need = done_cand > 0 && i + 1 < nch ? (int64_t)((double)end * (double)c->kp.ncand / (double)done_cand) + 1 : end;
            int rc = ensure_host_records(c, need > end ? need : end, base);
            if (rc) return rc;
            c->host_grows++;
        }
        if (m > 0) {
            stage_scope sc(c, SIFT3D_STAGE_DESCRIPTOR, 0.0, 0, m, ds);
            if (c->kp.p.sampler_cap > 0 && !(split && i > 0)) /* the per-CU tokens start from zero whatever became of an earlier launch
                                                                * (not under the first chunk's running kernel, which holds some) */
                HIPCHK(c, hipMemsetAsync(c->sampler_tokens, 0, sizeof(int) * SIFT3D_CU_SLOTS, ds));
            sift3d_kp_params q = c->kp.p;
            q.rec_shift = c->place.dst ? c->place.d_shift : nullptr;
            /* (with a shared destination the per-record group words still go to this context's own buffer, which then has to
             * hold them: ensure_kp_buffers sized it for the candidates, and a run that outgrows it falls back to growing it) */
            if (c->place.dst && end > c->hrecs_cap) {
                HIPCHK(c, hipStreamSynchronize(ds));
                int rc = ensure_host_records(c, end, 0);
                if (rc) return rc;
                c->host_grows++;
            }
            HIPCHK(c, sift3d_launch_descriptors(ds, q, c->kps, c->rec_kp + base, c->rec_frame + base, m,
                                                (c->place.dst ? c->place.dst : c->d_hrecs) + base, c->d_hgroup + base, c->kp.taps5));
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

int describe_finish(sift3d_ctx *c, int64_t *n_out)
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

#ifdef SIFT3D_DEV
/* Development builds only (tools/overlap_probe.py): would the keypoint kernel and the descriptor kernel gain from sharing the
 * CUs?  After an extraction (its sorted candidates, keypoints and record map still on the device) the two kernels are run
 * again, (a) one after the other as the pipeline does, (b) cut into slices of kslice extrema / dslice records queued
 * alternately on two streams, so that neither grid is ever large enough to keep the other out of the CUs.  The results are
 * the ones already there (same inputs); only the times matter.  out_ms: (a), (b). */
extern "C" int sift3d_dev_overlap_probe(sift3d_ctx *c, int kslice, int dslice, double *out_ms)
{
    if (!c || !out_ms || kslice < 1 || dslice < 1 || c->kp.ncand <= 0 || c->kp.nrec <= 0) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    float taps3[SIFT3D_MAX_TAPS];
    sift3d_gauss_taps(0.5f, 0.01f, taps3);
    const int64_t ncand = c->kp.ncand, nrec = c->kp.nrec;
    hipEvent_t ev[6];
    for (hipEvent_t &e : ev) HIPCHK(c, hipEventCreate(&e));
    hipStream_t s1 = c->stream, s2 = c->kp_stream;
    HIPCHK(c, hipStreamSynchronize(s1));
    HIPCHK(c, hipStreamSynchronize(s2));
    auto launch_k = [&](hipStream_t st, int64_t a, int64_t n) -> hipError_t {
        sift3d_kp_params q = c->kp.p;
        q.patch0 = c->patch0 + (size_t)a * SIFT3D_PATCH_VOX;
        return sift3d_launch_keypointsA(st, q, c->keys_b + a, c->vals_b + a, n, c->kps + a, c->nrec + a, taps3);
    };
    auto launch_d = [&](hipStream_t st, int64_t b, int64_t m) -> hipError_t {
        return sift3d_launch_descriptors(st, c->kp.p, c->kps, c->rec_kp + b, c->rec_frame + b, m, c->d_hrecs + b, c->d_hgroup + b, c->kp.taps5);
    };
    /* (a) */
    HIPCHK(c, hipEventRecord(ev[0], s1));
    HIPCHK(c, launch_k(s1, 0, ncand));
    HIPCHK(c, hipMemsetAsync(c->sampler_tokens, 0, sizeof(int) * SIFT3D_CU_SLOTS, s1));
    HIPCHK(c, launch_d(s1, 0, nrec));
    HIPCHK(c, hipEventRecord(ev[1], s1));
    HIPCHK(c, hipStreamSynchronize(s1));
    /* (b) */
    HIPCHK(c, hipMemsetAsync(c->sampler_tokens, 0, sizeof(int) * SIFT3D_CU_SLOTS, s1));
    HIPCHK(c, hipEventRecord(ev[2], s1));
    HIPCHK(c, hipStreamWaitEvent(s2, ev[2], 0));
    /* four streams for the slices of each kernel, so that the tail of one slice is covered by the next three */
    enum { NS = 4 };
    hipStream_t ks[NS], ds[NS];
    hipEvent_t done[2 * NS];
    for (int i = 0; i < NS; i++) {
        HIPCHK(c, hipStreamCreateWithFlags(&ks[i], hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&ds[i], hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&done[NS + i], hipEventDisableTiming));
        HIPCHK(c, hipStreamWaitEvent(ks[i], ev[2], 0));
        HIPCHK(c, hipStreamWaitEvent(ds[i], ev[2], 0));
    }
    (void)s2;
    int64_t a = 0, b = 0;
    int ki = 0, di = 0;
    while (a < ncand || b < nrec) { /* alternately, in proportion */
        if (a < ncand) {
            const int64_t n = ncand - a < kslice ? ncand - a : kslice;
            HIPCHK(c, launch_k(ks[ki++ % NS], a, n));
            a += n;
        }
        const int64_t b_to = ncand > 0 ? (int64_t)((double)nrec * (double)a / (double)ncand) : nrec;
        while (b < nrec && (b < b_to || a >= ncand)) {
            const int64_t m = nrec - b < dslice ? nrec - b : dslice;
            HIPCHK(c, launch_d(ds[di++ % NS], b, m));
            b += m;
        }
    }
    for (int i = 0; i < NS; i++) {
        HIPCHK(c, hipEventRecord(done[i], ks[i]));
        HIPCHK(c, hipEventRecord(done[NS + i], ds[i]));
        HIPCHK(c, hipStreamWaitEvent(s1, done[i], 0));
        HIPCHK(c, hipStreamWaitEvent(s1, done[NS + i], 0));
    }
    HIPCHK(c, hipEventRecord(ev[4], s1));
    HIPCHK(c, hipStreamSynchronize(s1));
    for (int i = 0; i < NS; i++) {
        hipStreamDestroy(ks[i]);
        hipStreamDestroy(ds[i]);
        hipEventDestroy(done[i]);
        hipEventDestroy(done[NS + i]);
    }
    float m0 = 0, m1 = 0;
    HIPCHK(c, hipEventElapsedTime(&m0, ev[0], ev[1]));
    HIPCHK(c, hipEventElapsedTime(&m1, ev[2], ev[4]));
    out_ms[0] = m0;
    out_ms[1] = m1;
    for (hipEvent_t e : ev) hipEventDestroy(e);
    return SIFT3D_OK;
}
#endif

static int describe_sorted(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand, int desc_mode,
                           float eig_thres, float size_factor, int64_t *n_out, bool levels_on_device = false)
{
    int rc = describe_queue(c, levels, ncand, desc_mode, eig_thres, size_factor, levels_on_device);
    if (!rc) rc = describe_launch(c);
    if (!rc) rc = describe_finish(c, n_out);
    return rc;
}

/* The end of run_pipeline when the candidate list is split (see there).  Called with everything queued: the first part's count is
 * on its way to h_split[0..2] behind ev_split[2] (kp_stream), the main stream ends behind both extrema streams.  *done = false:
 * something did not fit -- every stream has been drained and the extrema launches replayed into one list; the caller
 * continues with the one-list schedule. */
static int finish_split_tail(sift3d_ctx *c, const std::vector<sift3d_level> &levels, bool levels_on_device, int desc_mode, float eig_thres,
                             float size_factor, int64_t *n_out, bool *done)
{
    *done = false;
    unsigned long long *const h_split = c->h_cnt0 + 8 + SIFT3D_KP_MAX_CHUNKS;
    const int64_t capA = c->cand_split_at, capB = c->cand_cap - capA;
    hipStream_t ks = c->kp_stream;
    float taps3[SIFT3D_MAX_TAPS];
    /* overflow_mark: the own-level overflow word that triggered the fall-back (0: it was a validated-extrema list or a
     * per-keypoint buffer).  The remedy cand_finalize would apply after ANOTHER overflowing pass is applied here, before the
     * replay (advisor finding, round 4: three extrema passes instead of two on dense volumes). */
    auto fall_back = [&](unsigned long long overflow_mark) -> int {
        HIPCHK(c, hipStreamSynchronize(ks));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (overflow_mark > 0) {
            const int rc = surv_make_room(c, overflow_mark);
            if (rc) return rc;
        }
        return cand_replay(c); /* one list again; cand_finalize grows whatever else was too small */
    };
    /* the second part's count (and the first part's once more, with the overflow mark as it stands at the end) */
    h_split[3] = h_split[5] = h_split[7] = 0;
    HIPCHK(c, hipMemcpyAsync(h_split + 3, c->d_count, sizeof(unsigned long long) * 5, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev_split[2]));
    const int64_t nA = (int64_t)h_split[0];
    if (h_split[2] > 0 || nA > capA) return fall_back(h_split[2]);
    int rc = describe_begin(c, levels.size(), taps3);
    if (rc) return rc;
    /* room for the second part as well: it is rarely more than a fiftieth of the first */
    rc = ensure_kp_buffers(c, nA + nA / 8 + 4096, nA);
    if (rc) return rc;
    kp_params_of(c, desc_mode, eig_thres, size_factor, c->kp.p);
    c->kp.split = true;
    /* nothing else between the count and the sort: the level table went to the device on this stream when the run began, the
     * keypoint counter and the first chunk's record base were cleared with the extrema counters (cand_reset) */
    if (!levels_on_device)
        HIPCHK(c, hipMemcpyAsync(c->d_levels, levels.data(), sizeof(sift3d_level) * levels.size(), hipMemcpyHostToDevice, ks));
    int nch = 0;
    if (nA > 0) {
        HIPCHK(c, sift3d_sort_candidates(ks, c->sort_tmp, c->sort_tmp_bytes, c->keys_a, c->keys_b, c->vals_a, c->vals_b, nA));
        rc = describe_queue_chunk(c, nch++, 0, nA, ks, taps3);
        if (rc) return rc;
    }
    c->kp.nchunks = nch;
    /* the coarse octaves: the main stream holds nothing else */
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int64_t nB = (int64_t)h_split[7];
    if (h_split[5] > 0 || nB > capB || nA + nB > c->kps_cap || (int64_t)h_split[3] != nA) return fall_back(h_split[5]);
    if (nB > 0) {
        HIPCHK(c, sift3d_sort_candidates(ks, c->sort_tmp, c->sort_tmp_bytes, c->keys_a + capA, c->keys_b + nA, c->vals_a + capA, c->vals_b + nA, nB));
        rc = describe_queue_chunk(c, nch++, nA, nA + nB, ks, taps3);
        if (rc) return rc;
    }
    c->kp.nchunks = nch;
    c->kp.ncand = nA + nB;
    c->last.n_extrema = nA + nB;
    rc = describe_launch(c);
    if (!rc) rc = describe_finish(c, n_out);
    if (rc) return rc;
    *done = true;
    return SIFT3D_OK;
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
    /* Split tail (round 4).  Octaves 0 and 1 hold 98 - 99 % of the extrema and their detection passes are done while the chain of
     * small launches that builds the coarser octaves is still running (0.3 ms at 512^3, the chip mostly idle under it).  The
     * candidate list is therefore cut in two: [0, split) takes the extrema of octaves 0 and 1, the rest those of the coarser ones;
     * as soon as the first part's count is on the host it is sorted and its keypoint kernel starts on kp_stream, beside the
     * coarse chain; the second part follows as a second chunk of the per-keypoint stage.  The sort key leads with the level
     * id, so the two sorted parts back to back ARE the sorted whole: records and their order are unchanged.  Any overflow
     * (either part, the own-level lists, the per-keypoint buffers) falls back to the one-list schedule with a replay. */
    const bool split_tail = extract && oct.size() >= 3 && c->tune[SIFT3D_TUNE_SPLIT_TAIL] && c->timing != 3 && c->tune[SIFT3D_TUNE_KP_CHUNKS] == 0;
    if (split_tail) {
        int64_t second = c->cand_cap / 16 + 1024 < c->cand_cap / 2 ? c->cand_cap / 16 + 1024 : c->cand_cap / 2;
        if (c->tune[SIFT3D_TUNE_SPLIT_TAIL] == 2) second = 8; /* tests: the second part overflows, the fall-back runs with the first in flight */
        c->cand_split_at = c->cand_cap - second;
    }
    unsigned long long *const h_split = c->h_cnt0 + 8 + SIFT3D_KP_MAX_CHUNKS; /* pinned: [0..2] first part, [3..7] everything */

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
    /* split tail: the level table does not depend on anything the loop below finds out, so it goes to the device now, on the
     * stream the first part's keypoint kernel will run on, instead of between that part's count and its sort */
    std::vector<sift3d_level> levels_sent;
    bool levels_early_split = false;
    if (split_tail && levels.size() <= 96) {
        float sg[7] = {0, 0, 0, 0, 0, 0, 0}, s_ = 1.6f, fs = 1;
        sg[0] = s_;
        for (int j = 1; j < 6; j++) {
            s_ *= factor;
            sg[j] = s_;
        }
        for (size_t o = 0; o < oct.size(); o++) {
            for (int l = 0; l < 3; l++) {
                sift3d_level &lv = levels[o * 3 + (size_t)l];
                memset(&lv, 0, sizeof lv);
                lv.img = c->L[l + 1] + oct[o].off;
                lv.dogc = c->D[l + 1] + oct[o].off;
                lv.X = (int)oct[o].X; lv.Y = (int)oct[o].Y; lv.Z = (int)oct[o].Z;
                lv.XP = (int)oct[o].XP;
                lv.sigma_h = sg[l]; lv.sigma_c = sg[l + 1]; lv.sigma_l = sg[l + 2];
                lv.octave_factor = fs;
                lv.Zl = (int)oct[o].Z;
                lv.z_off = 0;
                lv.pad = 0;
            }
            fs *= 2.0f;
        }
        levels_sent = levels;
        HIPCHK(c, hipMemcpyAsync(c->d_levels, levels_sent.data(), sizeof(sift3d_level) * levels_sent.size(), hipMemcpyHostToDevice, c->kp_stream));
        levels_early_split = true;
    }
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
        c->cand_group = o >= 2 ? 1 : 0; /* only looked at while the list is split */
        c->surv_sel = (exs != c->stream && which == 1) ? 1 : 0;
        int rc_ = SIFT3D_OK;
        /* an octave one workgroup built whole (at most 4 096 voxels, every DoG level stored): its three detection levels in one
         * launch; the per-level jobs are recorded all the same, for a replay after a list overflow */
        const bool small_octave = pl.tiny_done && pl.d4tiny;
        if (small_octave) {
            const float *dl[5] = {c->D[0] + d.off, c->D[1] + d.off, c->D[2] + d.off, c->D[3] + d.off, pl.d4tiny};
            stage_scope sc(c, SIFT3D_STAGE_EXTREMA, 12.0 * (double)d.XP * d.Y * d.Z, 0, d.XP * d.Y * d.Z, exs);
            const cand_target tg = cand_target_of(c);
            HIPCHK(c, sift3d_launch_extrema_octave_small(exs, dl, d.XP, d.X, d.Y, d.Z, (int)o * 3, tg.keys, tg.vals, tg.count, tg.cap));
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
        c->cand_group = 0;
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
            bool sub_done = false;
            /* L_j = blur(L_{j-1}); D_{j-1} = L_{j-1} - L_j fused into the z pass */
            /* nothing reads L_5: only D_4 = L_4 - L_5 is needed, so the last level is not stored */
            if (!(lazy_next && j == 5)) {
                if (j == 5) {
                    rc = ensure_level_buffer(c, &c->D[4]);
                    if (rc) return rc;
                }
                float *dst_dog = (lazy && j == 1) ? nullptr : c->D[j - 1] + d.off;
                /* level 3 is what the next octave starts from: the launch that makes it writes the half-size volume too where it can */
                float *sub = nullptr;
                if (j == 3 && o + 1 < oct.size() && d.XP % 8 == 0 && oct[o + 1].XP == d.XP / 2) sub = c->L[0] + oct[o + 1].off;
                rc = blur_dev(c, c->L[j - 1] + d.off, j < 5 ? c->L[j] + d.off : nullptr, dst_dog, d.XP, d.Y, d.Z, ex, 0.01f, sub, &sub_done);
                if (rc) return rc;
                if (d.XP != d.X) /* the blur ran over the pitched width: its pad columns go back to zero */
                    HIPCHK(c, sift3d_launch_zero_pad(ws, j < 5 ? c->L[j] + d.off : nullptr, dst_dog, d.XP, d.X, d.Y * d.Z));
            }
            if (j == 3 && o + 1 < oct.size()) {
                if (!sub_done) {
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
            if (o == 1 && split_tail) { /* the first part is complete behind what the two extrema streams hold now */
                HIPCHK(c, hipEventRecord(c->ev_split[0], c->ex_stream));
                HIPCHK(c, hipStreamWaitEvent(c->kp_stream, c->ev_split[0], 0));
                if (used_second) {
                    HIPCHK(c, hipEventRecord(c->ev_split[1], c->ex_stream2));
                    HIPCHK(c, hipStreamWaitEvent(c->kp_stream, c->ev_split[1], 0));
                }
                h_split[0] = h_split[1] = h_split[2] = 0;
                HIPCHK(c, hipMemcpyAsync(h_split, c->d_count, sizeof(unsigned long long) * 3, hipMemcpyDeviceToHost, c->kp_stream));
                HIPCHK(c, hipEventRecord(c->ev_split[2], c->kp_stream));
            }
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
    if (split_tail) {
        bool done = false;
        /* the table the loop above filled in is the one that was uploaded before it (same expressions); checked, not assumed */
        const bool table_ok = levels_early_split && levels.size() == levels_sent.size() &&
                              memcmp(levels.data(), levels_sent.data(), sizeof(sift3d_level) * levels.size()) == 0;
        rc = finish_split_tail(c, levels, table_ok, desc_mode, eig_thres, size_factor, n_out, &done);
        if (rc) return rc;
        if (done) {
            *feats_out = c->h_recs; /* pinned, owned by the context */
            return SIFT3D_OK;
        }
        /* fell back: every stream is idle, the extrema launches were replayed into one list on the main stream */
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

/* sift3d_describe_dev in two halves, for a caller that places the records of several contexts -- the ranks of a Z-slab run, one
 * process each -- in ONE list (include/sift3d.h).  First half: everything up to and including the keypoint kernel, and this context's
 * records per (level, is_max) group. */
extern "C" int sift3d_describe_dev_counts(sift3d_ctx *c, const sift3d_level_desc *levels, int n_levels, int desc_mode, float eig_thres,
                                          float size_factor, const int32_t **group_counts, int64_t *n_records)
{
    if (!c || !group_counts || !n_records) return SIFT3D_ERR_ARG;
    if (desc_mode < SIFT3D_DESC_SIFT || desc_mode > SIFT3D_DESC_NRRIEF) return set_err(c, SIFT3D_ERR_ARG, "bad descriptor mode");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<sift3d_level> lv;
    int rc = levels_from_desc(c, levels, n_levels, lv);
    if (rc) return rc;
    rc = fence_in(c);
    if (rc) return rc;
    int64_t ncand = 0;
    rc = cand_finalize(c, &ncand);
    if (rc) return rc;
    c->last.n_extrema = ncand;
    describe_want_group_counts(c, true);
    const int chunks = c->tune[SIFT3D_TUNE_KP_CHUNKS];
    c->tune[SIFT3D_TUNE_KP_CHUNKS] = 1; /* the places need the whole list's counts before the one descriptor launch */
    rc = describe_queue(c, lv, ncand, desc_mode, eig_thres, size_factor, false);
    c->tune[SIFT3D_TUNE_KP_CHUNKS] = chunks;
    const int *hc = nullptr;
    if (!rc) rc = describe_group_counts(c, &hc, n_records);
    describe_want_group_counts(c, false);
    if (rc) return rc;
    *group_counts = hc;
    c->staged = 1;
    return SIFT3D_OK;
}

/* Second half: the descriptor kernel stores record i of group g at list[i + shift[g]]; list is host memory this context's device can
 * write (sift3d_host_register, or any pinned mapped allocation).  Ends with a host synchronisation. */
extern "C" int sift3d_describe_dev_place(sift3d_ctx *c, sift3d_feature *list, const int32_t *shift, const sift3d_feature **own_view,
                                         const int32_t **group_view, int64_t *n_out)
{
    if (!c || !n_out) return SIFT3D_ERR_ARG;
    if (!c->staged) return set_err(c, SIFT3D_ERR_ARG, "sift3d_describe_dev_place without sift3d_describe_dev_counts before it");
    c->staged = 0;
    HIPCHK(c, hipSetDevice(c->device));
    if (own_view) *own_view = nullptr;
    if (group_view) *group_view = nullptr;
    if (list && !shift) return set_err(c, SIFT3D_ERR_ARG, "sift3d_describe_dev_place: a list without its shifts");
    if (list && c->kp.ncand > 0 && c->kp.nchunks == 1) {
        sift3d_feature *dview = nullptr;
        HIPCHK(c, hipHostGetDevicePointer((void **)&dview, list, 0));
        int rc = describe_placement(c, dview, shift);
        if (rc) return rc;
    }
    int rc = describe_launch(c);
    if (!rc) rc = describe_finish(c, n_out);
    if (rc) return rc;
    if (own_view && !list) *own_view = c->h_recs; /* list == NULL: the records are where sift3d_describe_dev leaves them */
    if (group_view) *group_view = c->h_group;
    return SIFT3D_OK;
}

/* Host memory of the caller (e.g. a shared-memory segment every rank's process maps) made writable by every device of the process. */
extern "C" int sift3d_host_register(void *p, int64_t bytes)
{
    if (!p || bytes <= 0) return SIFT3D_ERR_ARG;
    return hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped) == hipSuccess ? SIFT3D_OK : SIFT3D_ERR_DEVICE;
}

extern "C" int sift3d_host_unregister(void *p)
{
    if (!p) return SIFT3D_ERR_ARG;
    return hipHostUnregister(p) == hipSuccess ? SIFT3D_OK : SIFT3D_ERR_DEVICE;
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

