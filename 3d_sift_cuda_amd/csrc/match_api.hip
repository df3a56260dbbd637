/*
 * match_api.hip -- C-ABI of the exact nearest-neighbour search (include/sift3d.h, "matcher").  The kernels are in
 * kernels_match.hip; the vote accumulation of the reference's matcher, which consumes these lists, is host code
 * (csrc/match_votes.c), as it is in the reference (R/feat_common/featMatchUtilities.cpp:1584-1819).
 */
#include <cstdio>
#include <cstring>

#include "sift3d_internal.h"

hipError_t sift3d_launch_knn_norms(hipStream_t s, const signed char *v, int64_t n, int *norms, unsigned long long *stats);
hipError_t sift3d_launch_knn(hipStream_t s, const signed char *db, const int *db_norm, int64_t n_db, const signed char *q, const int *q_norm,
                             int64_t n_q, int k, int const_norm, int groups, int segments, int *part_d, int *part_i, int *out_i, int *out_d);
int sift3d_knn_list_length(int k);
void sift3d_knn_plan(int64_t n_db, int64_t n_q, int k, int *groups, int *segments);

#define KCHK(call)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s failed: %s", #call, hipGetErrorString(e_)); \
            rc = SIFT3D_ERR_DEVICE;                                                                      \
            goto done;                                                                                   \
        }                                                                                                \
    } while (0)

extern "C" int sift3d_knn64(int device, const int8_t *db, int64_t n_db, const int8_t *queries, int64_t n_q, int k, int32_t *idx,
                            int32_t *dist2, int repeats, double *kernel_ms, char *err, int64_t err_len)
{
    int rc = SIFT3D_OK;
    signed char *d_db = nullptr, *d_q = nullptr;
    int *d_dbn = nullptr, *d_qn = nullptr, *d_pd = nullptr, *d_pi = nullptr, *d_oi = nullptr, *d_od = nullptr;
    unsigned long long *d_stats = nullptr, stats[6];
    int const_norm = -1;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (err && err_len > 0) err[0] = 0;
    if (kernel_ms) *kernel_ms = 0.0;
    const int KK = sift3d_knn_list_length(k);
    int groups = 1, segments = 1;
    /* row indices are 32-bit in the kernels, and the last tile is padded to a whole one: n_db + a tile must stay below 2^31,
     * or a pad row's index wraps and no longer compares >= n_db in the merge (advisor, round 3) */
    if (!db || !queries || !idx || !dist2 || n_db <= 0 || n_q <= 0 || k < 1 || KK == 0 || n_db > (1ll << 31) - 4096 || n_q > (1ll << 31) - 4096) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "bad arguments (1 <= k <= 32, 0 < n_db, n_q <= 2^31 - 4096)");
        return SIFT3D_ERR_ARG;
    }
    if (repeats < 1) repeats = 1;
    sift3d_knn_plan(n_db, n_q, k, &groups, &segments);
    KCHK(hipSetDevice(device));
    KCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    KCHK(hipEventCreate(&e0));
    KCHK(hipEventCreate(&e1));
    KCHK(hipMalloc((void **)&d_db, (size_t)n_db * 64));
    KCHK(hipMalloc((void **)&d_q, (size_t)n_q * 64));
    KCHK(hipMalloc((void **)&d_dbn, sizeof(int) * (size_t)n_db));
    KCHK(hipMalloc((void **)&d_qn, sizeof(int) * (size_t)n_q));
    KCHK(hipMalloc((void **)&d_pd, sizeof(int) * (size_t)n_q * 2 * segments * KK));
    KCHK(hipMalloc((void **)&d_pi, sizeof(int) * (size_t)n_q * 2 * segments * KK));
    KCHK(hipMalloc((void **)&d_oi, sizeof(int) * (size_t)n_q * k));
    KCHK(hipMalloc((void **)&d_od, sizeof(int) * (size_t)n_q * k));
    KCHK(hipMemcpyAsync(d_db, db, (size_t)n_db * 64, hipMemcpyHostToDevice, s));
    KCHK(hipMemcpyAsync(d_q, queries, (size_t)n_q * 64, hipMemcpyHostToDevice, s));
    /* The matrix cores take signed bytes: components must be 0..127 (rank descriptors are 0..63).  The norms kernel looks at
     * every byte anyway and reports the first offender -- and whether all database vectors have one squared norm: rank
     * descriptors do (every one a permutation of 0..63), and the search then needs no arithmetic on its candidates in the
     * common case (knn_search_kernel<KK, true>). */
    KCHK(hipMalloc((void **)&d_stats, sizeof(stats)));
    for (int it = 0; it < repeats; it++) { /* repeats > 1: timing (the first run is a warm-up; the last run's results are returned) */
        /* timing starts behind the warm-up run; with repeats == 1 there is none, and the one run's verdict on the bytes (a
         * device-to-host copy the host waits for) sits inside the interval: kernel_ms is then end to end, not kernel time */
        if (it == 0 || (it == 1 && repeats > 1)) KCHK(hipEventRecord(e0, s));
        stats[0] = stats[1] = stats[3] = stats[4] = ~0ull;
        stats[2] = stats[5] = 0;
        KCHK(hipMemcpyAsync(d_stats, stats, sizeof(stats), hipMemcpyHostToDevice, s));
        KCHK(sift3d_launch_knn_norms(s, d_db, n_db, d_dbn, d_stats));
        KCHK(sift3d_launch_knn_norms(s, d_q, n_q, d_qn, d_stats + 3));
        if (it == 0) { /* the verdict on the bytes: once (the timed repeats run the kernels again but do not wait for it) */
            KCHK(hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, s));
            KCHK(hipStreamSynchronize(s));
            if (stats[0] != ~0ull || stats[3] != ~0ull) {
                if (err && err_len > 0)
                    snprintf(err, (size_t)err_len, "%s component %llu is outside 0..127", stats[0] != ~0ull ? "database" : "query",
                             stats[0] != ~0ull ? stats[0] : stats[3]);
                rc = SIFT3D_ERR_ARG;
                goto done;
            }
            const_norm = stats[1] == stats[2] ? (int)stats[1] : -1;
        }
        KCHK(sift3d_launch_knn(s, d_db, d_dbn, n_db, d_q, d_qn, n_q, k, const_norm, groups, segments, d_pd, d_pi, d_oi, d_od));
    }
    KCHK(hipEventRecord(e1, s));
    KCHK(hipMemcpyAsync(idx, d_oi, sizeof(int) * (size_t)n_q * k, hipMemcpyDeviceToHost, s));
    KCHK(hipMemcpyAsync(dist2, d_od, sizeof(int) * (size_t)n_q * k, hipMemcpyDeviceToHost, s));
    KCHK(hipStreamSynchronize(s));
    if (kernel_ms) {
        float ms = 0;
        KCHK(hipEventElapsedTime(&ms, e0, e1));
        *kernel_ms = (double)ms / (repeats > 1 ? repeats - 1 : 1);
    }
done:
    hipFree(d_db); hipFree(d_q); hipFree(d_dbn); hipFree(d_qn); hipFree(d_pd); hipFree(d_pi); hipFree(d_oi); hipFree(d_od); hipFree(d_stats);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (s) hipStreamDestroy(s);
    return rc;
}
