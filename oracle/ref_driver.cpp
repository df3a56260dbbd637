// ref_driver.cpp -- C-linkage shims over the few reference translation units
// that compile from source as they lie under /root/reference (no CUDA, no
// znzlib needed): GaussianMask.cpp, PpImage.cpp, GenericImage.cpp,
// LocationValue.cpp and the header-only templates in SVD.h / MultiScale.h.
// Built only in the build container (never on the GPU box) into oracle/_ref/;
// used by tests/ to pin the corresponding oracle functions.  The shims hold
// no algorithm of their own.
#include <string.h>
#include "GaussianMask.h"
#include "LocationValue.h"
#include "MultiScale.h"
#include "PpImage.h"
#include "SVD.h"

extern "C" {

int ref_gauss_filter_size(float sigma, float min_value) { return calculate_gaussian_filter_size(sigma, min_value); }

// un-normalised taps of generate_gaussian_filter1d, called as
// gb3d_blur3d_interleave calls it (mean = n/2)
int ref_gauss_taps_raw(float sigma, int n, float *taps)
{
    PpImage img;
    img.Initialize(1, n, n * sizeof(float), sizeof(float) * 8);
    generate_gaussian_filter1d(img, sigma, n / 2);
    memcpy(taps, img.ImageRow(0), n * sizeof(float));
    return n;
}

void ref_svd3(float *mat9, float *w3, float *v9)
{
    float m[3][3], v[3][3], w[3];
    memcpy(m, mat9, sizeof(m));
    memset(v, 0, sizeof(v));
    memset(w, 0, sizeof(w));
    SingularValueDecomp<float, 3, 3>(m, w, v);
    memcpy(mat9, m, sizeof(m));
    memcpy(w3, w, sizeof(w));
    memcpy(v9, v, sizeof(v));
}

void ref_sort_eig(float *w3, float *v9)
{
    float v[3][3], w[3];
    memcpy(v, v9, sizeof(v));
    memcpy(w, w3, sizeof(w));
    SortEigenDecomp<float, 3>(w, v);
    memcpy(w3, w, sizeof(w));
    memcpy(v9, v, sizeof(v));
}

void ref_invert3(const float *in9, float *out9)
{
    float a[3][3], b[3][3];
    memcpy(a, in9, sizeof(a));
    invert_3x3<float, double>(a, b);
    memcpy(out9, b, sizeof(b));
}

void ref_mult3(const float *mat9, const float *in3, float *out3)
{
    float a[3][3], x[3], y[3];
    memcpy(a, mat9, sizeof(a));
    memcpy(x, in3, sizeof(x));
    mult_3x3<float, double>(a, x, y);
    memcpy(out3, y, sizeof(y));
}

// LOCATION_VALUE_XYZ is {int x,y,z; float fValue} == o3_extremum
void ref_sort_high_low(void *items, int n)
{
    LOCATION_VALUE_XYZ_ARRAY a;
    a.plvz = (LOCATION_VALUE_XYZ *)items;
    a.iCount = n;
    lvSortHighLow(a);
}

int ref_sign(float v) { return sign<float>(v); }
}
