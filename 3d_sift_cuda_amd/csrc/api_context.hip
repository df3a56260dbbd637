/*
 * api_context.hip -- the context of the C-ABI: device-resident pyramid buffers, candidate and per-keypoint buffers, stream and events, tuning knobs, the LDS self-test
 *
 * One of the five translation units behind include/sift3d.h (round 6: api.hip, 2 300 lines, cut at its seams; no behaviour
 * change): api_context.hip (contexts, buffers, tuning, stream), api_timing.hip (event pairs, the launch log), api_ops.hip
 * (the blur dispatcher, the operator-level entry points, the candidate lists), api_pipeline.hip (volume upload, the
 * per-keypoint stage, run_pipeline, sift3d_extract / sift3d_detect), api_slab.hip (the building blocks a Z-slab driver calls).
 * What they share is pipeline.h.  R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/
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
std::vector<octave_dims> octave_list(int64_t X, int64_t Y, int64_t Z)
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
int alloc_cands(sift3d_ctx *c, int64_t cap)
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

/* T[0], T[1] hold at least `floats` floats each (see ctx_create).  Growing waits for the stream: a blur queued earlier may
 * still be using the old pair. */
int ensure_T(sift3d_ctx *c, int64_t floats)
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
