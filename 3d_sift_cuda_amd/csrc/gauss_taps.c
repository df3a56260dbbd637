/*
 * gauss_taps.c -- host-side Gaussian tap generation for the blur kernels.
 *
 * The taps decide keypoints at the last bit, so they are produced on the host
 * with the same libm calls, types and order as the reference:
 * calculate_gaussian_filter_size and generate_gaussian_filter1d
 * (R/src_common/GaussianMask.cpp:12-57, 241-265; the reference is C++, where
 * exp() of a float is expf()) followed by the float normalisation of
 * gb3d_blur3d_interleave (R/src_common/GaussBlur3D.cpp:1190-1201).
 * sift3d_set_libm_variant(SIFT3D_LIBM_GCC5) switches exp() to what the
 * toolchain of the reference's shipped CPU binary made of it: the C
 * exp(double), the tap's product formed in double (include/sift3d.h).
 * R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/
 */
#include <math.h>

#include "sift3d.h"

#define SIFT3D_MAX_TAPS 129

static int libm_variant = SIFT3D_LIBM_CURRENT;

int sift3d_set_libm_variant(int which)
{
    if (which != SIFT3D_LIBM_CURRENT && which != SIFT3D_LIBM_GCC5) return SIFT3D_ERR_ARG;
    const int before = libm_variant;
    libm_variant = which;
    return before;
}

int sift3d_get_libm_variant(void) { return libm_variant; }

/* exp() of a float as the selected build of the reference evaluates it, narrowed to float as the reference's casts do */
static float ref_exp(float x) { return libm_variant == SIFT3D_LIBM_GCC5 ? (float)exp((double)x) : expf(x); }

static int filter_size(float sigma, float min_value)
{
    if (sigma == 0) return 1;
    float value = ref_exp(0.0f);
    float cur = 1, nxt = 1, power;
    int i = 0;
    do {
        i++;
        cur = nxt;
        power = ((float)(i * i)) / ((float)-2.0 * sigma * sigma);
        nxt = cur + 2 * ref_exp(power);
    } while (nxt - cur > 0.00001f);
    for (i = 1; value <= cur * (1.0f - min_value); i++) {
        power = ((float)(i * i)) / ((float)-2.0 * sigma * sigma);
        value += 2 * ref_exp(power);
    }
    i--;
    return 2 * i + 1;
}

int sift3d_gauss_taps(float sigma, float min_value, float *taps)
{
    if (!(sigma >= 0.0f) || !(min_value >= 0.0f && min_value < 1.0f) || !taps) return SIFT3D_ERR_ARG;
    int n = filter_size(sigma, min_value);
    if (n > SIFT3D_MAX_TAPS) return SIFT3D_ERR_ARG;
    if (sigma > 0.0f) {
        const double pi = 3.1415926535897932384626433832795;
        const float mean = (float)(n / 2);
        const float sig2 = sigma * sigma;
        const float scale = (float)(1.0 / (sigma * sqrt(2.0 * pi)));
        for (int j = 0; j < n; j++) {
            float pos = ((float)j - mean);
            float power = ((pos * pos) / sig2) / (float)(-2.0f);
            if (libm_variant == SIFT3D_LIBM_GCC5) taps[j] = (float)((double)scale * exp((double)power));
            else taps[j] = (float)(scale * expf(power));
        }
    } else {
        taps[0] = 1;
    }
    float sum = 0;
    for (int c = 0; c < n; c++) sum += taps[c];
    for (int c = 0; c < n; c++) taps[c] /= sum;
    return n;
}
