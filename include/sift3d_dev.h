/*
 * sift3d_dev.h -- what is NOT part of the drop-in boundary of include/sift3d.h.
 *
 * (1) One hardware self-test that the product library exports, because the GPU test suite runs it on the library that
 *     ships: it checks the one hardware property an implementation choice rests on.
 * (2) Development hooks that exist only in `make DEV=1` builds (3d_sift_cuda_amd/csrc/_build_dev: the timing-ablation
 *     branches of the per-keypoint kernels, matcher plans, the overlap probe).  The product library does not export them
 *     (tests/test_abi_and_host.py checks), and nothing under 3d_sift_cuda_amd/ or tests/ calls them: tools/ only.
 * A maintainer of the reference needs neither.
 */
#ifndef SIFT3D_DEV_H
#define SIFT3D_DEV_H
#include "sift3d.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Hardware self-test behind one implementation choice (no reference counterpart): the orientation-histogram
 * splat accumulates with LDS float atomics (ds_add_f32), which is only a drop-in for the reference's float
 * additions if the LDS unit rounds exactly like the vector ALU.  For i < n: valu[i] = a[i] + b[i] on the vector
 * ALU, lds[i] = the same sum through ds_add_f32 (host arrays).  tests/ compare the two bit for bit on random,
 * denormal, signed-zero, infinite and NaN operands. */
int sift3d_selftest_lds_add(sift3d_ctx *ctx, const float *a, const float *b, int64_t n, float *valu, float *lds);

#ifdef SIFT3D_DEV
/* Development builds only (make DEV=1; tools/kp_ablate.py, tools/desc_ablate.py): the per-keypoint kernels return after
 * stage n (0 = run everything).  Not compiled into the product library. */
int sift3d_dev_set_stop(sift3d_ctx *ctx, int n);
/* Development builds only (tools/bench_match.py with KNN_PLAN=groups,segments): overrides how sift3d_knn64 cuts a search
 * (0 = the library's own choice). */
void sift3d_dev_knn_plan(int groups, int segments);
/* Development builds only (KNN_AHEAD=2 tools/bench_match.py): the search kernel with the matrix cores two subtiles ahead of the
 * vector unit (three accumulator sets) instead of one. */
void sift3d_dev_knn_ahead(int ahead);
/* Development builds only (tools/overlap_probe.py): the keypoint and the descriptor kernel of the last extraction run again,
 * one after the other (out_ms[0]) and in alternating slices on two streams (out_ms[1]). */
int sift3d_dev_overlap_probe(sift3d_ctx *ctx, int kslice, int dslice, double *out_ms);
#endif

#ifdef __cplusplus
}
#endif
#endif
