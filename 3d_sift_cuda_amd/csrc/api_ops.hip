/*
 * api_ops.hip -- the blur dispatcher (fused march or three passes), the operator-level entry points that mirror the reference's four accelerator wrappers, and the candidate lists of a run (reset / append / count / finalize)
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

int check_shape(sift3d_ctx *c, int64_t nx, int64_t ny, int64_t nz)
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
int fence_in(sift3d_ctx *c)
{
    if (!c->own_stream) return SIFT3D_OK;
    HIPCHK(c, hipEventRecord(c->ev_fence[0], nullptr));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_fence[0], 0));
    return SIFT3D_OK;
}

int fence_out(sift3d_ctx *c)
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

int cand_replay(sift3d_ctx *c)
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
int surv_make_room(sift3d_ctx *c, unsigned long long high_water)
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
