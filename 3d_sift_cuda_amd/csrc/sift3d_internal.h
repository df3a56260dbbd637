/*
 * sift3d_internal.h -- declarations shared by the HIP translation units of
 * libsift3d_hip.so (kernel launchers and the context).  Not installed.
 */
#ifndef SIFT3D_INTERNAL_H
#define SIFT3D_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sift3d.h"
#include "sift3d_dev.h" /* the self-test and, in DEV builds, the development hooks: declared apart from the boundary */

#define SIFT3D_MAX_TAPS 129
#define SIFT3D_FAST_MAX_R 8 /* templated kernels cover 3..17 taps (every sigma the pyramid uses) */
#define SIFT3D_PATCH_DIM 11
#define SIFT3D_PATCH_VOX 1331

struct sift3d_taps {
    float f[2 * SIFT3D_FAST_MAX_R + 1];
};

/* Candidates are kept as (key, value) pairs so that one device radix sort puts them in the
 * reference's order: key = level id << 40 | is_max << 39 | linear voxel index, with
 * level id = octave*3 + (DoG level - 1).  Minima sort before maxima, raster order inside. */
#define SIFT3D_KEY_LVL_SHIFT 40
#define SIFT3D_KEY_MAX_SHIFT 39
#define SIFT3D_KEY_IDX_MASK ((1ull << 39) - 1ull)
/* an own-level extremum on its way to the two-level validation */
struct sift3d_survivor {
    long long idx;
    float value;
    int is_max;
};
/* an extremum that passed the test against the level below, on its way to the test against a level above that is
 * evaluated around it instead of being stored (extrema_validate_lazy_kernel) */
struct sift3d_survivor2 {
    int x, y, z;   /* position in the (pitched) volume */
    int is_max;
    float value;
    float h;       /* the level below at the extremum */
};
struct sift3d_cval {
    float value, h, l, pad; /* DoG at the extremum, one level below (H), one level above (L) */
};

/* one detection level of one octave (table in device memory) */
struct sift3d_level {
    const float *img;  /* Gaussian level L_k the keypoints are sampled from */
    const float *dogc; /* DoG level k (centre) */
    int X, Y, Z;       /* dims of the whole octave volume (Z is the global slice count) */
    int XP;            /* row pitch of img/dogc in floats (== X for a dense volume) */
    float sigma_h, sigma_c, sigma_l;
    float octave_factor; /* 2^octave */
    int Zl;    /* slices held in img/dogc (== Z on one GPU; slab + halos in Z-slab mode) */
    int z_off; /* global z of local slice 0 */
    int pad;
};

/* ---- kernel launchers (kernels_volume.hip) ---- */
hipError_t sift3d_launch_blur_x(hipStream_t s, const float *in, float *out, int64_t X, int64_t Y, int64_t Z,
                                const float *taps, int ntaps, const float *d_taps);
hipError_t sift3d_launch_blur_y(hipStream_t s, const float *in, float *out, int64_t X, int64_t Y, int64_t Z,
                                const float *taps, int ntaps, const float *d_taps);
/* prev/dog may be NULL (no DoG epilogue) */
hipError_t sift3d_launch_blur_z(hipStream_t s, const float *in, float *out, const float *prev, float *dog, int64_t X,
                                int64_t Y, int64_t Z, const float *taps, int ntaps, const float *d_taps);
/* all three passes and the DoG in one kernel; hipErrorNotSupported when the shape is outside it.  tune (may be NULL):
 * what sift3d_set_tuning forces -- 0 = the launcher's own choice */
struct sift3d_blur_tuning {
    int z_chunks;        /* SIFT3D_TUNE_FUSED_CHUNKS */
    int rows_per_thread; /* SIFT3D_TUNE_FUSED_ROWS: 2 = 512 threads, two planes of prefetch; 1 = 1024 threads, one plane */
    int tile;            /* SIFT3D_TUNE_FUSED_TILE: 1 = 64 x 32, 2 = 128 x 16 (two-rows-per-thread mapping, up to 13 taps) */
    int order;           /* SIFT3D_TUNE_FUSED_ORDER: which workgroup takes which tile (0 = by measurement, 1 .. 3: see the kernel) */
    int stagger;         /* SIFT3D_TUNE_FUSED_STAGGER: the second half of a workgroup's wavefronts half a step behind the first (0 = by measurement, 1 = off, 2 = on) */
};
hipError_t sift3d_launch_blur_fused(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z,
                                    const float *taps, int ntaps, const sift3d_blur_tuning *tune, int64_t zo0 = 0, int64_t zo1 = -1,
                                    float *sub = nullptr, int *sub_done = nullptr);
hipError_t sift3d_launch_dog(hipStream_t s, const float *a, const float *b, float *out, int64_t n);
hipError_t sift3d_launch_subsample(hipStream_t s, const float *in, int64_t X, int64_t Xl, int64_t Y, int64_t Z, float *out,
                                   int64_t XPout);
hipError_t sift3d_launch_zero_pad(hipStream_t s, float *a, float *b, int64_t X, int64_t Xl, int64_t rows);
/* levels 1..5 (L[4] may be NULL) and DoGs 0..4 of an octave of at most SIFT3D_TINY_VOX voxels from its level 0, rows of
 * pitch XP; hipErrorNotSupported outside that */
#define SIFT3D_TINY_VOX 4096
/* contexts of at most this many floats allocate the two pass intermediates of the three-launch blur when they are created */
#define SIFT3D_EAGER_T_FLOATS (1ll << 31)
struct sift3d_octave_out {
    float *L[5];
    float *D[5];
};
struct sift3d_octave_taps {
    float f[5][2 * SIFT3D_FAST_MAX_R + 1];
    int n[5];
};
hipError_t sift3d_launch_tiny_octave(hipStream_t s, const float *L0, const sift3d_octave_out &o, int64_t X, int64_t XP, int64_t Y,
                                     int64_t Z, const sift3d_octave_taps &t);
hipError_t sift3d_launch_double_size(hipStream_t s, const float *in, int64_t X, int64_t Y, int64_t Z, float *out);
hipError_t sift3d_launch_halve_size(hipStream_t s, const float *in, int64_t X, int64_t Y, int64_t Z, float *out);
/* X: row pitch, Xl: logical row length (Xl == X for a dense volume) */
/* Neighbour levels that are not stored as DoG volumes (NULL or all-zero: both neighbours are the stored dprev / dnext).
 * prev_b: the level below is dprev - prev_b, two Gaussian levels.  next_g: the level above is next_g - blur(next_g, taps),
 * and blur(next_g) is evaluated only at the 27 voxels around each extremum that passed everything else (dnext is ignored;
 * ntaps <= 2 * SIFT3D_FAST_MAX_R + 1, rows of whole 16-byte vectors). */
struct sift3d_extrema_lazy {
    const float *prev_b;
    const float *next_g;
    float taps[2 * SIFT3D_FAST_MAX_R + 1];
    int ntaps;
    sift3d_survivor2 *list2;         /* 64 segments of list2_cap / 64 entries, one per slab of z, like the own-level list */
    unsigned long long *list2_count; /* SIFT3D_LIST2_COUNTERS words, zeroed by the caller */
    int64_t list2_cap;               /* at least surv_cap of the same call */
};
hipError_t sift3d_launch_extrema(hipStream_t s, const float *dprev, const float *dcur, const float *dnext, int64_t X,
                                 int64_t Xl, int64_t Y, int64_t Z, int z_lo, int z_hi, int lvl_id, unsigned long long *keys,
                                 sift3d_cval *vals, unsigned long long *count, int64_t cap, sift3d_survivor *surv,
                                 unsigned long long *surv_count /* SIFT3D_SURV_COUNTERS words */,
                                 unsigned long long *surv_overflow, int64_t surv_cap, bool zero_counters,
                                 const sift3d_extrema_lazy *lazy = nullptr);
/* the three detection levels of an octave of at most SIFT3D_TINY_VOX voxels (d[0..4]: its five stored DoG levels) in one launch */
hipError_t sift3d_launch_extrema_octave_small(hipStream_t s, const float *const d[5], int64_t X, int64_t Xl, int64_t Y, int64_t Z,
                                              int lvl_id0, unsigned long long *keys, sift3d_cval *vals, unsigned long long *count,
                                              int64_t cap);
#define SIFT3D_SURV_SETS 96 /* one counter set per extrema pass of a pipeline run, zeroed together */
#define SIFT3D_SURV_COUNTERS (64 * 32)
#define SIFT3D_LIST2_COUNTERS 64

/* ---- per-keypoint stage (kernels_keypoint.hip) ---- */
struct sift3d_kp_params {
    const sift3d_level *levels; /* device table indexed by level id */
    float eig_thres;
    float size_factor;
    int desc_mode;
    int debug_stop; /* -DSIFT3D_DEV builds only (timing ablation, tools/kp_ablate.py): the kernels return after stage N; 0 = run everything */
    float *patch0;  /* per extremum: the identity-frame patch (1331 floats, normalised once) that phase A sampled anyway;
                     * phase B reads it for the un-reoriented record instead of sampling it again */
    int *sampler_tokens; /* phase B: per-CU count of workgroups in their sampling phase (SIFT3D_CU_SLOTS ints, zero between runs) */
    int sampler_cap;     /* at most this many per CU sample at a time (0: no limit) */
    int desc_seg;        /* descriptor kernel: records per segment of the XCD-contiguous order (a multiple of 8; 0: the whole list is one) */
    const int *rec_shift; /* descriptor kernel: NULL, or SIFT3D_GROUPS ints -- record r of group g is stored at slot r + rec_shift[g] (several
                           * contexts writing one merged list: the slab driver) */
};
/* SIFT3D_GROUPS (include/sift3d.h, 193): level id * 2 + is_max for up to 96 levels, and one slot for anything beyond */
hipError_t sift3d_launch_group_counts(hipStream_t s, const unsigned long long *keys, const int *nrec, int64_t ncand, int *counts);
#define SIFT3D_CU_SLOTS 2048
#define SIFT3D_MAX_FRAMES 11 /* determineCanonicalOrientation3D stops at FEATURE_3D_DIM frames */
/* phase A result per extremum */
struct sift3d_dkp {
    float x, y, z, scale; /* octave coordinates, +0.5 applied */
    float eigs[3];
    float ori0[9]; /* sorted eigenvectors (record 0) */
    int nframes;
    float frames[SIFT3D_MAX_FRAMES * 9];
    int lvl;
    unsigned info;
};
/* nrec[k] = 0 (rejected) or 1 + number of canonical frames */
hipError_t sift3d_launch_keypointsA(hipStream_t s, const sift3d_kp_params &p, const unsigned long long *keys,
                                    const sift3d_cval *vals, int64_t ncand, sift3d_dkp *kps, int *nrec, const float *taps3);
/* nrec / offs: one chunk of the candidate list starting at candidate cand_off; rec_base[0] = its first record (in), rec_base[1] =
 * the first record of the next chunk (out) */
hipError_t sift3d_launch_recmap(hipStream_t s, const int *nrec, const int *offs, int64_t ncand, int cand_off, int *rec_base,
                                int *rec_kp, int *rec_frame, unsigned long long *kp_count);
hipError_t sift3d_launch_descriptors(hipStream_t s, const sift3d_kp_params &p, const sift3d_dkp *kps, const int *rec_kp,
                                     const int *rec_frame, int64_t nrec, sift3d_feature *recs, int *rec_group,
                                     const float *taps5);
/* device sort / scan (sort_scan.hip, rocPRIM) */
size_t sift3d_sort_temp_bytes(int64_t n);
hipError_t sift3d_sort_candidates(hipStream_t s, void *temp, size_t temp_bytes, const unsigned long long *keys_in,
                                  unsigned long long *keys_out, const sift3d_cval *vals_in, sift3d_cval *vals_out, int64_t n);
size_t sift3d_scan_temp_bytes(int64_t n);
hipError_t sift3d_scan_counts(hipStream_t s, void *temp, size_t temp_bytes, const int *in, int *out, int64_t n);

#endif
