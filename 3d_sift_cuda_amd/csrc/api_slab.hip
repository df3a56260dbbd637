/*
 * api_slab.hip -- the building blocks a Z-slab driver calls on a slab context: candidate lists over caller-owned level buffers, the lazy level-above test, the per-keypoint stage on a level table, host registration of a shared record list
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
