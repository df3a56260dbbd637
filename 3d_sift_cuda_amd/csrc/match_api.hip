/*
 * match_api.hip -- C-ABI of the exact nearest-neighbour search (include/sift3d.h, "matcher").  The kernels are in
 * kernels_match.hip; the vote accumulation of the reference's matcher, which consumes these lists, is host code
 * (csrc/match_votes.c), as it is in the reference (R/feat_common/featMatchUtilities.cpp:1584-1819).
 */
#include <cstdio>
#include <cstring>

#include "sift3d_internal.h"

hipError_t sift3d_launch_knn_norms(hipStream_t s, const signed char *v, int64_t n, int *norms);
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
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (err && err_len > 0) err[0] = 0;
    if (kernel_ms) *kernel_ms = 0.0;
    const int KK = sift3d_knn_list_length(k);
    int groups = 1, segments = 1;
    if (!db || !queries || !idx || !dist2 || n_db <= 0 || n_q <= 0 || k < 1 || KK == 0 || n_db >= (1ll << 31)) {
        if (err && err_len > 0) snprintf(err, (size_t)err_len, "bad arguments (1 <= k <= 32, 0 < n_db < 2^31)");
        return SIFT3D_ERR_ARG;
    }
    /* the matrix cores take signed bytes: components must be 0..127 (rank descriptors are 0..63).  While looking at every
     * byte anyway: do all database vectors have one squared norm?  Rank descriptors do (every one a permutation of 0..63),
     * and the search then needs no arithmetic on its candidates in the common case (knn_search_kernel<KK, true>). */
    int const_norm = -1;
    for (int64_t v = 0; v < n_db; v++) {
        int nrm = 0;
        for (int j = 0; j < 64; j++) {
            const int c = db[v * 64 + j];
            if (c < 0) {
                if (err && err_len > 0) snprintf(err, (size_t)err_len, "database component %lld is outside 0..127", (long long)(v * 64 + j));
                return SIFT3D_ERR_ARG;
            }
            nrm += c * c;
        }
        if (v == 0) const_norm = nrm;
        else if (nrm != const_norm) const_norm = -2;
    }
    if (const_norm < 0) const_norm = -1;
    for (int64_t i = 0; i < n_q * 64; i++)
        if (queries[i] < 0) {
            if (err && err_len > 0) snprintf(err, (size_t)err_len, "query component %lld is outside 0..127", (long long)i);
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
    for (int it = 0; it < repeats; it++) { /* repeats > 1: timing (the last run's results are returned) */
        if (it == repeats - 1 || it == 0) KCHK(hipEventRecord(it == 0 ? e0 : e1, s));
        if (it == 0 && repeats > 1) { /* the first run is a warm-up; time the rest */ }
        KCHK(sift3d_launch_knn_norms(s, d_db, n_db, d_dbn));
        KCHK(sift3d_launch_knn_norms(s, d_q, n_q, d_qn));
        KCHK(sift3d_launch_knn(s, d_db, d_dbn, n_db, d_q, d_qn, n_q, k, const_norm, groups, segments, d_pd, d_pi, d_oi, d_od));
        if (it == 0 && repeats > 1) KCHK(hipEventRecord(e0, s)); /* timing starts behind the warm-up run */
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
    hipFree(d_db); hipFree(d_q); hipFree(d_dbn); hipFree(d_qn); hipFree(d_pd); hipFree(d_pi); hipFree(d_oi); hipFree(d_od);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (s) hipStreamDestroy(s);
    return rc;
}
