// ref_driver.cpp -- C-linkage shims over the few reference translation units
// that compile from source as they lie under /root/reference (no CUDA, no
// znzlib needed): GaussianMask.cpp, PpImage.cpp, GenericImage.cpp,
// LocationValue.cpp, PpImageFloatOutput.cpp and the header-only templates in
// SVD.h / MultiScale.h (3x3 algebra, the .key reader and writers, DistSqrPCs).
// Built only in the build container (never on the GPU box) into oracle/_ref/;
// used by tests/ to pin the corresponding oracle functions.  The shims hold
// no algorithm of their own.
#include <string.h>
#include "GaussianMask.h"
#include "LocationValue.h"
#include "MultiScale.h"
#include "PpImage.h"
#include "PpImageFloatOutput.h"
#include "SVD.h"
#include <stdint.h>
#include <stdlib.h>
#include <vector>

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

// ---- the .key reader / writers: header templates over FEATURE_TYPE that touch data members only
// (MultiScale.h:228-303 binary writer, :305-384 text reader, :386-474 text writer).  Feature3DInfo itself cannot be the
// argument: its constructor lives in MultiScale.cpp, which does not compile here.  RefFeat carries the member names the
// templates use and nothing else; the record the tests pass in and get back is the product's sift3d_feature layout
// {x, y, z, scale, ori[9], eigs[3], info, desc[64]} (324 bytes), copied member by member.
struct RefFeat {
    unsigned int m_uiInfo;
    float x, y, z, scale;
    float ori[3][3];
    float eigs[3];
    float m_pfPC[Feature3DInfo::FEATURE_3D_PCS];
};
struct PlainRec {
    float x, y, z, scale, ori[9], eigs[3];
    uint32_t info;
    float desc[64];
};

static void to_ref(const PlainRec *in, int n, std::vector<RefFeat> &v)
{
    v.resize(n);
    for (int i = 0; i < n; i++) {
        v[i].m_uiInfo = in[i].info;
        v[i].x = in[i].x; v[i].y = in[i].y; v[i].z = in[i].z; v[i].scale = in[i].scale;
        memcpy(v[i].ori, in[i].ori, sizeof(v[i].ori));
        memcpy(v[i].eigs, in[i].eigs, sizeof(v[i].eigs));
        memcpy(v[i].m_pfPC, in[i].desc, sizeof(v[i].m_pfPC));
    }
}

int ref_write_key_text(const void *recs, int n, const char *path, float eig_thres, int n_comments, char **comments)
{
    std::vector<RefFeat> v;
    to_ref((const PlainRec *)recs, n, v);
    return msFeature3DVectorOutputText<RefFeat>(v, (char *)path, eig_thres, n_comments, comments);
}

int ref_write_key_bin(const void *recs, int n, const char *path, float eig_thres)
{
    std::vector<RefFeat> v;
    to_ref((const PlainRec *)recs, n, v);
    return msFeature3DVectorOutputBin<RefFeat>(v, (char *)path, eig_thres);
}

// returns the template's return code; *n_out = records read (at most cap are copied out)
int ref_read_key_text(const char *path, void *recs, int cap, int *n_out)
{
    std::vector<RefFeat> v;
    int rc = msFeature3DVectorInputText<RefFeat>(v, (char *)path, -1);
    *n_out = 0;
    if (rc != 0) return rc;
    PlainRec *out = (PlainRec *)recs;
    *n_out = (int)v.size();
    for (int i = 0; i < (int)v.size() && i < cap; i++) {
        out[i].info = v[i].m_uiInfo;
        out[i].x = v[i].x; out[i].y = v[i].y; out[i].z = v[i].z; out[i].scale = v[i].scale;
        memcpy(out[i].ori, v[i].ori, sizeof(v[i].ori));
        memcpy(out[i].eigs, v[i].eigs, sizeof(v[i].eigs));
        memcpy(out[i].desc, v[i].m_pfPC, sizeof(v[i].m_pfPC));
    }
    return 0;
}

// Feature3DInfo::DistSqrPCs (MultiScale.h:61-73), the matcher's distance: an inline member that reads m_pfPC only, called
// on storage of the class's size (no constructor runs: it is not linkable here, and the member needs none)
float ref_dist_sqr_pcs(const float *a64, const float *b64, int n_pcs)
{
    Feature3DInfo *fa = (Feature3DInfo *)calloc(1, sizeof(Feature3DInfo));
    Feature3DInfo *fb = (Feature3DInfo *)calloc(1, sizeof(Feature3DInfo));
    memcpy(fa->m_pfPC, a64, sizeof(fa->m_pfPC));
    memcpy(fb->m_pfPC, b64, sizeof(fb->m_pfPC));
    float d = fa->DistSqrPCs(*fb, n_pcs);
    free(fa);
    free(fb);
    return d;
}

// the rotation of a record's orientation frame into world coordinates, the three calls of featExtract.cpp:535-537
void ref_world_orientation(const float *rot9, float *ori9)
{
    float rot[3][3], ori[3][3], inv[3][3], out[3][3];
    memcpy(rot, rot9, sizeof(rot));
    memcpy(ori, ori9, sizeof(ori));
    invert_3x3<float, float>(ori, inv);
    mult_3x3_matrix<float, float>(rot, inv, out);
    invert_3x3<float, float>(out, ori);
    memcpy(ori9, ori, sizeof(ori));
}

// image.pgm: output_float(PpImage of floats, file name) (PpImageFloatOutput.cpp:131-180, GenericImage::WriteToFile), called
// as MultiScale.cpp:373-384 does on an x-y slice: rows = y, cols = x, row step = x floats
int ref_output_float_pgm(const float *slice, int rows, int cols, const char *path)
{
    PpImage img;
    if (!img.InitializeSubImage(rows, cols, cols * sizeof(float), sizeof(float) * 8, (unsigned char *)slice)) return -1;
    output_float(img, (char *)path);
    return 0;
}
}
