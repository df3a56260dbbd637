/*
 * gauss_taps.c -- host-side Gaussian tap generation for the blur kernels.
 *
 * The taps decide keypoints at the last bit, so they are produced on the host
 * with the same libm calls, types and order as the reference:
 * calculate_gaussian_filter_size and generate_gaussian_filter1d
 * (R/src_common/GaussianMask.cpp:12-57, 241-265; the reference is C++, where
 * exp() of a float is expf()) followed by the float normalisation of
 * gb3d_blur3d_interleave (R/src_common/GaussBlur3D.cpp:1190-1201).
 * R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/
 */
#include <math.h>

#include "sift3d.h"

#define SIFT3D_MAX_TAPS 129

static int filter_size(float sigma, float min_value)
{
    if (sigma == 0) return 1;
    float value = expf(0.0f);
    float cur = 1, nxt = 1, power;
    int i = 0;
    do {
        i++;
        cur = nxt;
        power = ((float)(i * i)) / ((float)-2.0 * sigma * sigma);
        nxt = cur + 2 * expf(power);
    } while (nxt - cur > 0.00001f);
    for (i = 1; value <= cur * (1.0f - min_value); i++) {
        power = ((float)(i * i)) / ((float)-2.0 * sigma * sigma);
        value += 2 * expf(power);
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
            taps[j] = (float)(scale * expf(power));
        }
    } else {
        taps[0] = 1;
    }
    float sum = 0;
    for (int c = 0; c < n; c++) sum += taps[c];
    for (int c = 0; c < n; c++) taps[c] /= sum;
    return n;
}
