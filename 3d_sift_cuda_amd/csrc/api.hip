/*
 * api.hip -- the C-ABI of include/sift3d.h: context (device-resident pyramid
 * buffers, stream, timing events), operator-level entry points and the
 * scale-space / extraction pipeline that strings the kernels together.
 *
 * Schedule = msGeneratePyramidDOG3D_efficient (R/src_common/MultiScale.cpp:236-570,
 * R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/): initial
 * blur to sigma 1.6, then per octave five blurs L1..L5 (sigma ratio 2^(1/3)),
 * DoG k = L_k - L_{k+1} for k = 0..4, extrema in DoG 1..3, keypoints sampled
 * from L_k, next octave seeded by the 2x2x2 mean of L_3.  The reference
 * recycles five buffers and validates "on the fly"; here all six Gaussians
 * and five DoGs of an octave stay resident in HBM (13.1 N floats in total:
 * 288 GB holds a 1024^3 volume four times over), so each level is produced by
 * exactly one x, one y and one z(+DoG) pass and read by one extrema pass.
 */
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sift3d_internal.h"

struct timed_launch {
    int stage;
    hipEvent_t e0, e1;
    int ntaps;
    int64_t nvox;
    double bytes;
    float ms;
};

struct sift3d_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int64_t capN;
    float *vol;   /* input volume */
    float *L[6];  /* Gaussian levels */
    float *D[5];  /* DoG levels */
    float *T[2];  /* x- and y-pass intermediates */
    float *half;  /* next octave seed (capN/8) */
    float *L0_full, *half_small; /* the two allocations L[0] and half alternate between */
    float *d_taps;
    sift3d_dcand *cand;
    int64_t cand_cap; /* per level segment */
    unsigned long long *d_counts;
    sift3d_dkp *kps;
    int64_t kps_cap;
    int *rec_kp, *rec_frame;
    sift3d_feature *recs;
    int64_t recs_cap;
    int64_t nx, ny, nz;
    bool has_volume;
    bool timing;
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

extern "C" int sift3d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void sift3d_free(void *p) { free(p); }

extern "C" const char *sift3d_last_error(const sift3d_ctx *ctx) { return ctx ? ctx->err : "no context"; }

/* L[0] and half swap allocations from octave to octave; every entry point starts from the original roles */
static void roles_reset(sift3d_ctx *c)
{
    c->L[0] = c->L0_full;
    c->half = c->half_small;
}

static void free_dev(sift3d_ctx *c)
{
    roles_reset(c);
    hipFree(c->vol);
    for (int i = 0; i < 6; i++) hipFree(c->L[i]);
    for (int i = 0; i < 5; i++) hipFree(c->D[i]);
    hipFree(c->T[0]);
    hipFree(c->T[1]);
    hipFree(c->half);
    hipFree(c->d_taps);
    hipFree(c->cand);
    hipFree(c->d_counts);
    hipFree(c->kps);
    hipFree(c->rec_kp);
    hipFree(c->rec_frame);
    hipFree(c->recs);
}

extern "C" sift3d_ctx *sift3d_create(int device, int64_t nx, int64_t ny, int64_t nz)
{
    if (nx <= 0 || ny <= 0 || nz <= 0) return nullptr;
    int n = sift3d_device_count();
    if (device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    sift3d_ctx *c = new sift3d_ctx();
    c->device = device;
    c->own_stream = true;
    c->capN = nx * ny * nz;
    c->err[0] = 0;
    c->timing = false;
    c->pool_used = 0;
    c->resolved = 0;
    c->has_volume = false;
    c->nx = c->ny = c->nz = 0;
    memset(&c->last, 0, sizeof(c->last));
    c->vol = nullptr;
    for (int i = 0; i < 6; i++) c->L[i] = nullptr;
    for (int i = 0; i < 5; i++) c->D[i] = nullptr;
    c->T[0] = c->T[1] = c->half = c->d_taps = nullptr;
    c->L0_full = c->half_small = nullptr;
    c->cand = nullptr;
    c->d_counts = nullptr;
    c->kps = nullptr;
    c->rec_kp = c->rec_frame = nullptr;
    c->recs = nullptr;
    c->kps_cap = c->recs_cap = 0;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    const size_t vb = sizeof(float) * (size_t)c->capN;
    ok = ok && hipMalloc((void **)&c->vol, vb) == hipSuccess;
    for (int i = 0; i < 6 && ok; i++) ok = hipMalloc((void **)&c->L[i], vb) == hipSuccess;
    for (int i = 0; i < 5 && ok; i++) ok = hipMalloc((void **)&c->D[i], vb) == hipSuccess;
    for (int i = 0; i < 2 && ok; i++) ok = hipMalloc((void **)&c->T[i], vb) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->half, vb / 8 + 64) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_taps, sizeof(float) * SIFT3D_MAX_TAPS) == hipSuccess;
    c->cand_cap = c->capN / 64 + 4096;
    ok = ok && hipMalloc((void **)&c->cand, sizeof(sift3d_dcand) * (size_t)c->cand_cap * 3) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_counts, sizeof(unsigned long long) * 4) == hipSuccess;
    c->L0_full = c->L[0];
    c->half_small = c->half;
    if (!ok) {
        free_dev(c);
        if (c->stream) hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void sift3d_destroy(sift3d_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    free_dev(c);
    for (hipEvent_t e : c->pool) hipEventDestroy(e);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

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

extern "C" int sift3d_sync(sift3d_ctx *c)
{
    if (!c) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    stage_scope(sift3d_ctx *c_, int stage_, double bytes_, int ntaps_ = 0, int64_t nvox_ = 0)
        : c(c_), stage(stage_), e0(nullptr), e1(nullptr), ntaps(ntaps_), nvox(nvox_), bytes(bytes_)
    {
        c->last.launches[stage] += 1;
        c->last.alg_bytes[stage] += bytes;
        if (c->timing) {
            e0 = get_event(c);
            e1 = get_event(c);
            hipEventRecord(e0, c->stream);
        }
    }
    ~stage_scope()
    {
        if (c->timing) {
            hipEventRecord(e1, c->stream);
            c->launches.push_back({stage, e0, e1, ntaps, nvox, bytes, 0.0f});
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
    c->timing = on != 0;
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
    }
    return *n > cap ? SIFT3D_ERR_CAPACITY : SIFT3D_OK;
}

/* ---- device-level building blocks -------------------------------------- */
/* out = blur(in); if dog != NULL also dog = in - out.  Uses T[0], T[1]. */
static int blur_dev(sift3d_ctx *c, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, float sigma,
                    float min_value)
{
    float taps[SIFT3D_MAX_TAPS];
    int n = sift3d_gauss_taps(sigma, min_value, taps);
    if (n < 0) return set_err(c, SIFT3D_ERR_ARG, "bad blur parameters sigma=%g min=%g", sigma, min_value);
    const double N = (double)X * Y * Z;
    if (n == 1) { /* delta filter: out = 1*in */
        HIPCHK(c, hipMemcpyAsync(out, in, sizeof(float) * (size_t)N, hipMemcpyDeviceToDevice, c->stream));
        if (dog) HIPCHK(c, hipMemsetAsync(dog, 0, sizeof(float) * (size_t)N, c->stream));
        return SIFT3D_OK;
    }
    if (n / 2 > SIFT3D_FAST_MAX_R)
        HIPCHK(c, hipMemcpyAsync(c->d_taps, taps, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    {
        stage_scope sc(c, SIFT3D_STAGE_BLUR_X, 8.0 * N, n, (int64_t)N);
        HIPCHK(c, sift3d_launch_blur_x(c->stream, in, c->T[0], X, Y, Z, taps, n, c->d_taps));
    }
    {
        stage_scope sc(c, SIFT3D_STAGE_BLUR_Y, 8.0 * N, n, (int64_t)N);
        HIPCHK(c, sift3d_launch_blur_y(c->stream, c->T[0], c->T[1], X, Y, Z, taps, n, c->d_taps));
    }
    {
        stage_scope sc(c, SIFT3D_STAGE_BLUR_Z_DOG, (dog ? 16.0 : 8.0) * N, n, (int64_t)N);
        HIPCHK(c, sift3d_launch_blur_z(c->stream, c->T[1], out, dog ? in : nullptr, dog, X, Y, Z, taps, n, c->d_taps));
    }
    return SIFT3D_OK;
}

static int check_shape(sift3d_ctx *c, int64_t nx, int64_t ny, int64_t nz)
{
    if (!c) return SIFT3D_ERR_ARG;
    roles_reset(c);
    if (nx <= 0 || ny <= 0 || nz <= 0 || nx * ny * nz > c->capN)
        return set_err(c, SIFT3D_ERR_ARG, "volume %lldx%lldx%lld does not fit the context (%lld voxels)", (long long)nx,
                       (long long)ny, (long long)nz, (long long)c->capN);
    if (nx >= (1ll << 31) || ny >= (1ll << 31) || nz >= 65536 + 2) return set_err(c, SIFT3D_ERR_ARG, "dimension too large");
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
    return blur_dev(c, d_in, d_out, nullptr, nx, ny, nz, sigma, min_value);
}

extern "C" int sift3d_gauss_blur_dog_dev(sift3d_ctx *c, const float *d_in, float *d_out, float *d_dog, int64_t nx,
                                         int64_t ny, int64_t nz, float sigma, float min_value)
{
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (nz < 2) return set_err(c, SIFT3D_ERR_ARG, "2-D images are outside this path (featExtract rejects z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    return blur_dev(c, d_in, d_out, d_dog, nx, ny, nz, sigma, min_value);
}

extern "C" int sift3d_gauss_blur(sift3d_ctx *c, const float *in, float *out, int64_t nx, int64_t ny, int64_t nz,
                                 float sigma, float min_value)
{
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
    return SIFT3D_OK;
}

extern "C" int sift3d_dog_dev(sift3d_ctx *c, const float *d_a, const float *d_b, float *d_out, int64_t n)
{
    if (!c || n <= 0) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sift3d_launch_dog(c->stream, d_a, d_b, d_out, n));
    return SIFT3D_OK;
}

extern "C" int sift3d_dog(sift3d_ctx *c, const float *a, const float *b, float *out, int64_t n)
{
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
    return SIFT3D_OK;
}

extern "C" int sift3d_subsample2_dev(sift3d_ctx *c, const float *d_in, int64_t nx, int64_t ny, int64_t nz, float *d_out)
{
    if (!c || nx < 2 || ny < 2 || nz < 2) return set_err(c, SIFT3D_ERR_ARG, "subsample needs every dimension >= 2");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sift3d_launch_subsample(c->stream, d_in, nx, ny, nz, d_out));
    return SIFT3D_OK;
}

extern "C" int sift3d_subsample2(sift3d_ctx *c, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out)
{
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
    return SIFT3D_OK;
}

extern "C" int sift3d_double_size(sift3d_ctx *c, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out)
{
    if (!c || !in || !out || nx < 2 || ny < 2 || nz < 2 || 8 * nx * ny * nz > c->capN)
        return set_err(c, SIFT3D_ERR_ARG, "double_size: the context must hold the doubled volume");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->L[0], in, sizeof(float) * (size_t)(nx * ny * nz), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, sift3d_launch_double_size(c->stream, c->L[0], nx, ny, nz, c->L[1]));
    HIPCHK(c, hipMemcpyAsync(out, c->L[1], sizeof(float) * (size_t)(8 * nx * ny * nz), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_volume = false;
    return SIFT3D_OK;
}

extern "C" int sift3d_halve_size(sift3d_ctx *c, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out)
{
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
    return SIFT3D_OK;
}

/* Runs the extrema kernel for one level into candidate segment `seg`, grows
 * the buffer and reruns on overflow, and returns the raster-ordered list
 * (minima first, then maxima, each by linear index) in host memory. */
static bool cand_less(const sift3d_dcand &a, const sift3d_dcand &b)
{
    if (a.is_max != b.is_max) return a.is_max < b.is_max;
    return a.idx < b.idx;
}

static int grow_cands(sift3d_ctx *c, int64_t need)
{
    int64_t ncap = need + need / 2 + 4096;
    sift3d_dcand *n = nullptr;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMalloc((void **)&n, sizeof(sift3d_dcand) * (size_t)ncap * 3));
    hipFree(c->cand);
    c->cand = n;
    c->cand_cap = ncap;
    return SIFT3D_OK;
}

static int extrema_levels(sift3d_ctx *c, const float *const *dp, const float *const *dc, const float *const *dn, int nlev,
                          int64_t X, int64_t Y, int64_t Z, std::vector<sift3d_dcand> *out /* nlev vectors */)
{
    for (int attempt = 0; attempt < 3; attempt++) {
        HIPCHK(c, hipMemsetAsync(c->d_counts, 0, sizeof(unsigned long long) * 4, c->stream));
        for (int l = 0; l < nlev; l++) {
            stage_scope sc(c, SIFT3D_STAGE_EXTREMA, 4.0 * (double)X * Y * Z, 0, X * Y * Z);
            HIPCHK(c, sift3d_launch_extrema(c->stream, dp[l], dc[l], dn[l], X, Y, Z, c->cand + (size_t)l * c->cand_cap,
                                            c->d_counts + l, c->cand_cap));
        }
        unsigned long long cnt[4] = {0, 0, 0, 0};
        HIPCHK(c, hipMemcpyAsync(cnt, c->d_counts, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int64_t mx = 0;
        for (int l = 0; l < nlev; l++) mx = std::max<int64_t>(mx, (int64_t)cnt[l]);
        if (mx > c->cand_cap) {
            int rc = grow_cands(c, mx);
            if (rc) return rc;
            /* the relaunch repeats the work; drop this attempt's launch counts */
            continue;
        }
        for (int l = 0; l < nlev; l++) {
            out[l].resize((size_t)cnt[l]);
            if (cnt[l])
                HIPCHK(c, hipMemcpyAsync(out[l].data(), c->cand + (size_t)l * c->cand_cap, sizeof(sift3d_dcand) * cnt[l],
                                         hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int l = 0; l < nlev; l++) std::sort(out[l].begin(), out[l].end(), cand_less);
        return SIFT3D_OK;
    }
    return set_err(c, SIFT3D_ERR_MEMORY, "candidate buffer could not be grown");
}

extern "C" int sift3d_extrema(sift3d_ctx *c, const float *d_prev, const float *d_cur, const float *d_next, int64_t nx,
                              int64_t ny, int64_t nz, sift3d_extremum *minima, int64_t cap_min, int64_t *n_min,
                              sift3d_extremum *maxima, int64_t cap_max, int64_t *n_max)
{
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!d_prev || !d_cur || !n_min || !n_max) return set_err(c, SIFT3D_ERR_ARG, "null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t b = sizeof(float) * (size_t)(nx * ny * nz);
    HIPCHK(c, hipMemcpyAsync(c->D[0], d_prev, b, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->D[1], d_cur, b, hipMemcpyHostToDevice, c->stream));
    if (d_next) HIPCHK(c, hipMemcpyAsync(c->D[2], d_next, b, hipMemcpyHostToDevice, c->stream));
    c->has_volume = false;
    const float *dp[1] = {c->D[0]}, *dc[1] = {c->D[1]}, *dn[1] = {d_next ? c->D[2] : nullptr};
    std::vector<sift3d_dcand> v[1];
    rc = extrema_levels(c, dp, dc, dn, 1, nx, ny, nz, v);
    if (rc) return rc;
    int64_t a = 0, m = 0;
    bool over = false;
    for (const sift3d_dcand &d : v[0]) {
        sift3d_extremum e;
        e.x = (int32_t)(d.idx % nx);
        e.y = (int32_t)((d.idx / nx) % ny);
        e.z = (int32_t)(d.idx / (nx * ny));
        e.value = d.value;
        if (d.is_max) {
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
extern "C" int sift3d_set_volume(sift3d_ctx *c, const float *vol, int64_t nx, int64_t ny, int64_t nz)
{
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!vol) return set_err(c, SIFT3D_ERR_ARG, "null volume");
    if (nz <= 1) return set_err(c, SIFT3D_ERR_ARG, "Could not read volume (z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->vol, vol, sizeof(float) * (size_t)(nx * ny * nz), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nx = nx; c->ny = ny; c->nz = nz;
    c->has_volume = true;
    return SIFT3D_OK;
}

extern "C" int sift3d_set_volume_dev(sift3d_ctx *c, const float *d_vol, int64_t nx, int64_t ny, int64_t nz)
{
    int rc = check_shape(c, nx, ny, nz);
    if (rc) return rc;
    if (!d_vol) return set_err(c, SIFT3D_ERR_ARG, "null volume");
    if (nz <= 1) return set_err(c, SIFT3D_ERR_ARG, "Could not read volume (z <= 1)");
    HIPCHK(c, hipSetDevice(c->device));
    if (d_vol != c->vol)
        HIPCHK(c, hipMemcpyAsync(c->vol, d_vol, sizeof(float) * (size_t)(nx * ny * nz), hipMemcpyDeviceToDevice, c->stream));
    c->nx = nx; c->ny = ny; c->nz = nz;
    c->has_volume = true;
    return SIFT3D_OK;
}

static int ensure_kp_buffers(sift3d_ctx *c, int64_t ncand, int64_t nrec)
{
    if (ncand > c->kps_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->kps);
        c->kps = nullptr;
        c->kps_cap = ncand + ncand / 2 + 1024;
        HIPCHK(c, hipMalloc((void **)&c->kps, sizeof(sift3d_dkp) * (size_t)c->kps_cap));
    }
    if (nrec > c->recs_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->recs);
        hipFree(c->rec_kp);
        hipFree(c->rec_frame);
        c->recs = nullptr;
        c->rec_kp = c->rec_frame = nullptr;
        c->recs_cap = nrec + nrec / 2 + 1024;
        HIPCHK(c, hipMalloc((void **)&c->recs, sizeof(sift3d_feature) * (size_t)c->recs_cap));
        HIPCHK(c, hipMalloc((void **)&c->rec_kp, sizeof(int) * (size_t)c->recs_cap));
        HIPCHK(c, hipMalloc((void **)&c->rec_frame, sizeof(int) * (size_t)c->recs_cap));
    }
    return SIFT3D_OK;
}

static int run_pipeline(sift3d_ctx *c, float init_scale, bool extract, int desc_mode, float eig_thres, float size_factor,
                        std::vector<sift3d_candidate> *cands_out, std::vector<sift3d_feature> *feats_out)
{
    if (!c) return SIFT3D_ERR_ARG;
    if (!c->has_volume) return set_err(c, SIFT3D_ERR_ARG, "no volume set (sift3d_set_volume)");
    HIPCHK(c, hipSetDevice(c->device));
    roles_reset(c);
    timing_begin(c);
    int64_t X = c->nx, Y = c->ny, Z = c->nz;

    /* sigma schedule, MultiScale.cpp:288-294,369,526-527 (float arithmetic as there) */
    float sigma_init = 0.5f;
    if (init_scale > 0) sigma_init /= init_scale;
    float sigma = 1.6f;
    const float factor = (float)pow(2.0, 1.0 / (double)3);
    const float extra0 = sqrtf(sigma * sigma - sigma_init * sigma_init);
    float taps3[SIFT3D_MAX_TAPS], taps5[SIFT3D_MAX_TAPS];
    if (sift3d_gauss_taps(0.5f, 0.01f, taps3) != 3 || sift3d_gauss_taps((float)0.95, (float)0.01, taps5) != 5)
        return set_err(c, SIFT3D_ERR_ARG, "unexpected patch tap counts");

    int rc = blur_dev(c, c->vol, c->L[0], nullptr, X, Y, Z, extra0, 0.01f);
    if (rc) return rc;

    float fscale = 1;
    float sig[7];
    std::vector<sift3d_dkp> h_kps;
    std::vector<int> h_rec_kp, h_rec_frame;
    for (int oct = 0;; oct++) {
        sigma = 1.6f;
        sig[0] = sigma;
        if (X <= 2 || Y <= 2 || Z <= 2) break;
        const double N = (double)X * Y * Z;
        for (int j = 1; j < 6; j++) {
            const float ex = sigma * sqrtf(factor * factor - 1.0f);
            /* L_j = blur(L_{j-1}); D_{j-1} = L_{j-1} - L_j fused into the z pass */
            rc = blur_dev(c, c->L[j - 1], c->L[j], c->D[j - 1], X, Y, Z, ex, 0.01f);
            if (rc) return rc;
            if (j == 3) {
                stage_scope sc(c, SIFT3D_STAGE_SUBSAMPLE, 4.5 * N, 0, (int64_t)N);
                HIPCHK(c, sift3d_launch_subsample(c->stream, c->L[3], X, Y, Z, c->half));
            }
            sigma *= factor;
            sig[j] = sigma;
        }
        /* extrema of DoG 1..3 against their neighbours in scale */
        const float *dp[3] = {c->D[0], c->D[1], c->D[2]};
        const float *dc[3] = {c->D[1], c->D[2], c->D[3]};
        const float *dn[3] = {c->D[2], c->D[3], c->D[4]};
        std::vector<sift3d_dcand> lv[3];
        rc = extrema_levels(c, dp, dc, dn, 3, X, Y, Z, lv);
        if (rc) return rc;
        for (int l = 0; l < 3; l++) {
            c->last.n_extrema += (int64_t)lv[l].size();
            if (cands_out) {
                for (const sift3d_dcand &d : lv[l]) {
                    sift3d_candidate o;
                    o.octave = oct;
                    o.level = l + 1;
                    o.is_max = d.is_max;
                    o.x = (int32_t)(d.idx % X);
                    o.y = (int32_t)((d.idx / X) % Y);
                    o.z = (int32_t)(d.idx / (X * Y));
                    o.value = d.value;
                    o.h_value = d.h;
                    o.l_value = d.l;
                    cands_out->push_back(o);
                }
            }
            if (extract && !lv[l].empty()) {
                const int64_t nc = (int64_t)lv[l].size();
                rc = ensure_kp_buffers(c, nc, 0);
                if (rc) return rc;
                sift3d_dcand *seg = c->cand + (size_t)l * c->cand_cap;
                HIPCHK(c, hipMemcpyAsync(seg, lv[l].data(), sizeof(sift3d_dcand) * (size_t)nc, hipMemcpyHostToDevice, c->stream));
                sift3d_kp_params p;
                p.img = c->L[l + 1];
                p.dogc = c->D[l + 1];
                p.X = (int)X; p.Y = (int)Y; p.Z = (int)Z;
                p.sigma_h = sig[l]; p.sigma_c = sig[l + 1]; p.sigma_l = sig[l + 2];
                p.eig_thres = eig_thres;
                p.octave_factor = fscale;
                p.size_factor = size_factor;
                p.desc_mode = desc_mode;
                p.debug_stop = getenv("SIFT3D_KP_STOP") ? atoi(getenv("SIFT3D_KP_STOP")) : 0;
                {
                    stage_scope sc(c, SIFT3D_STAGE_KEYPOINT, 0.0, 0, nc);
                    HIPCHK(c, sift3d_launch_keypointsA(c->stream, p, seg, nc, c->kps, taps3));
                }
                h_kps.resize((size_t)nc);
                HIPCHK(c, hipMemcpyAsync(h_kps.data(), c->kps, sizeof(sift3d_dkp) * (size_t)nc, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                h_rec_kp.clear();
                h_rec_frame.clear();
                for (int64_t k = 0; k < nc; k++) {
                    if (h_kps[k].nrec <= 0) continue;
                    c->last.n_keypoints++;
                    for (int f = -1; f < h_kps[k].nframes; f++) {
                        h_rec_kp.push_back((int)k);
                        h_rec_frame.push_back(f);
                    }
                }
                const int64_t nr = (int64_t)h_rec_kp.size();
                if (nr > 0) {
                    rc = ensure_kp_buffers(c, 0, nr);
                    if (rc) return rc;
                    HIPCHK(c, hipMemcpyAsync(c->rec_kp, h_rec_kp.data(), sizeof(int) * (size_t)nr, hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(c->rec_frame, h_rec_frame.data(), sizeof(int) * (size_t)nr, hipMemcpyHostToDevice, c->stream));
                    {
                        stage_scope sc(c, SIFT3D_STAGE_DESCRIPTOR, 0.0, 0, nr);
                        HIPCHK(c, sift3d_launch_descriptors(c->stream, p, c->kps, c->rec_kp, c->rec_frame, nr, c->recs, taps5));
                    }
                    const size_t at = feats_out->size();
                    feats_out->resize(at + (size_t)nr);
                    HIPCHK(c, hipMemcpyAsync(feats_out->data() + at, c->recs, sizeof(sift3d_feature) * (size_t)nr, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                }
            }
        }
        fscale *= 2.0f;
        X /= 2; Y /= 2; Z /= 2;
        std::swap(c->L[0], c->half); /* the half buffer holds N/8 floats: enough for every later octave */
        c->last.n_octaves++;
    }
    roles_reset(c);
    timing_end(c);
    if (feats_out) c->last.n_records = (int64_t)feats_out->size();
    return SIFT3D_OK;
}

extern "C" int sift3d_detect(sift3d_ctx *c, float initial_image_scale, sift3d_candidate **out, int64_t *n_out)
{
    if (!c || !out || !n_out) return SIFT3D_ERR_ARG;
    std::vector<sift3d_candidate> v;
    int rc = run_pipeline(c, initial_image_scale, false, 0, 140.0f, 1.0f, &v, nullptr);
    if (rc) return rc;
    *n_out = (int64_t)v.size();
    *out = (sift3d_candidate *)malloc(sizeof(sift3d_candidate) * (v.size() ? v.size() : 1));
    if (!*out) return set_err(c, SIFT3D_ERR_MEMORY, "out of host memory");
    if (!v.empty()) memcpy(*out, v.data(), sizeof(sift3d_candidate) * v.size());
    return SIFT3D_OK;
}

extern "C" int sift3d_extract(sift3d_ctx *c, float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                              sift3d_feature **out, int64_t *n_out)
{
    if (!c || !out || !n_out) return SIFT3D_ERR_ARG;
    if (desc_mode < SIFT3D_DESC_SIFT || desc_mode > SIFT3D_DESC_NRRIEF) return set_err(c, SIFT3D_ERR_ARG, "bad descriptor mode");
    std::vector<sift3d_feature> v;
    int rc = run_pipeline(c, initial_image_scale, true, desc_mode, eig_thres, size_factor, nullptr, &v);
    if (rc) return rc;
    *n_out = (int64_t)v.size();
    *out = (sift3d_feature *)malloc(sizeof(sift3d_feature) * (v.size() ? v.size() : 1));
    if (!*out) return set_err(c, SIFT3D_ERR_MEMORY, "out of host memory");
    if (!v.empty()) memcpy(*out, v.data(), sizeof(sift3d_feature) * v.size());
    return SIFT3D_OK;
}
