/*
 * api_pipeline.hip -- the scale-space / extraction pipeline that strings the kernels
 * together: volume upload, the per-keypoint stage, run_pipeline, sift3d_extract /
 * sift3d_detect (round 6: one of the five translation units api.hip was cut into).
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
int candidates_to_host(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand,
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

int describe_sorted(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand, int desc_mode,
                           float eig_thres, float size_factor, int64_t *n_out, bool levels_on_device)
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
