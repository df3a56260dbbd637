/*
 * sift3d_internal.h -- declarations shared by the HIP translation units of
 * libsift3d_hip.so (kernel launchers and the context).  Not installed.
 */
#ifndef SIFT3D_INTERNAL_H
#define SIFT3D_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sift3d.h"

#define SIFT3D_MAX_TAPS 129
#define SIFT3D_FAST_MAX_R 8 /* templated kernels cover 3..17 taps (every sigma the pyramid uses) */
#define SIFT3D_PATCH_DIM 11
#define SIFT3D_PATCH_VOX 1331

struct sift3d_taps {
    float f[2 * SIFT3D_FAST_MAX_R + 1];
};

/* device-side candidate record written by the extrema kernel */
struct sift3d_dcand {
    long long idx; /* linear voxel index in the octave volume */
    float value, h, l;
    int is_max;
};

/* ---- kernel launchers (kernels_volume.hip) ---- */
hipError_t sift3d_launch_blur_x(hipStream_t s, const float *in, float *out, int64_t X, int64_t Y, int64_t Z,
                                const float *taps, int ntaps, const float *d_taps);
hipError_t sift3d_launch_blur_y(hipStream_t s, const float *in, float *out, int64_t X, int64_t Y, int64_t Z,
                                const float *taps, int ntaps, const float *d_taps);
/* prev/dog may be NULL (no DoG epilogue) */
hipError_t sift3d_launch_blur_z(hipStream_t s, const float *in, float *out, const float *prev, float *dog, int64_t X,
                                int64_t Y, int64_t Z, const float *taps, int ntaps, const float *d_taps);
hipError_t sift3d_launch_dog(hipStream_t s, const float *a, const float *b, float *out, int64_t n);
hipError_t sift3d_launch_subsample(hipStream_t s, const float *in, int64_t X, int64_t Y, int64_t Z, float *out);
hipError_t sift3d_launch_double_size(hipStream_t s, const float *in, int64_t X, int64_t Y, int64_t Z, float *out);
hipError_t sift3d_launch_halve_size(hipStream_t s, const float *in, int64_t X, int64_t Y, int64_t Z, float *out);
hipError_t sift3d_launch_extrema(hipStream_t s, const float *dprev, const float *dcur, const float *dnext, int64_t X,
                                 int64_t Y, int64_t Z, sift3d_dcand *out, unsigned long long *count, int64_t cap);

/* ---- per-keypoint stage (kernels_keypoint.hip) ---- */
struct sift3d_kp_params {
    const float *img;  /* Gaussian level L_k of the octave */
    const float *dogc; /* DoG level k (centre) */
    int X, Y, Z;
    float sigma_h, sigma_c, sigma_l;
    float eig_thres;
    float octave_factor; /* 2^octave */
    float size_factor;
    int desc_mode;
    int debug_stop; /* development aid: phase A returns after stage N (0 = run everything) */
};
#define SIFT3D_MAX_FRAMES 11 /* determineCanonicalOrientation3D stops at FEATURE_3D_DIM frames */
#define SIFT3D_RECS_PER_KP (1 + SIFT3D_MAX_FRAMES)
/* phase A result per extremum */
struct sift3d_dkp {
    float x, y, z, scale; /* octave coordinates, +0.5 applied */
    float eigs[3];
    float ori0[9]; /* sorted eigenvectors (record 0) */
    int nframes;
    float frames[SIFT3D_MAX_FRAMES * 9];
    int nrec; /* 0 = rejected, else 1 + nframes */
    unsigned info;
};
hipError_t sift3d_launch_keypointsA(hipStream_t s, const sift3d_kp_params &p, const sift3d_dcand *cands, int64_t ncand,
                                    sift3d_dkp *kps, const float *taps3);
hipError_t sift3d_launch_descriptors(hipStream_t s, const sift3d_kp_params &p, const sift3d_dkp *kps, const int *rec_kp,
                                     const int *rec_frame, int64_t nrec, sift3d_feature *recs, const float *taps5);

#endif
