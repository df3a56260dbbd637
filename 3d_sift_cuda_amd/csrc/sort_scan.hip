/*
 * sort_scan.hip -- the two library primitives of the pipeline (rocPRIM): a
 * radix sort of the extrema (key, value) pairs, which puts them in the order
 * the reference's serial scan produces (octave, level, minima before maxima,
 * raster index: R/src_common/MultiScale.cpp:2332-2398, 1361-1421), and an
 * exclusive scan of the per-keypoint record counts.
 */
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "sift3d_internal.h"

static const unsigned SORT_END_BIT = 48; /* 39 index bits + 1 + 7 level bits, rounded up */

size_t sift3d_sort_temp_bytes(int64_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const float4 *)nullptr, (float4 *)nullptr, (size_t)n, 0, SORT_END_BIT, (hipStream_t)0);
    return bytes;
}

hipError_t sift3d_sort_candidates(hipStream_t s, void *temp, size_t temp_bytes, const unsigned long long *keys_in,
                                  unsigned long long *keys_out, const sift3d_cval *vals_in, sift3d_cval *vals_out, int64_t n)
{
    if (n <= 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, (const float4 *)vals_in, (float4 *)vals_out,
                                     (size_t)n, 0, SORT_END_BIT, s);
}

size_t sift3d_scan_temp_bytes(int64_t n)
{
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const int *)nullptr, (int *)nullptr, 0, (size_t)n, rocprim::plus<int>(),
                                  (hipStream_t)0);
    return bytes;
}

hipError_t sift3d_scan_counts(hipStream_t s, void *temp, size_t temp_bytes, const int *in, int *out, int64_t n)
{
    if (n <= 0) return hipSuccess;
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), s);
}
