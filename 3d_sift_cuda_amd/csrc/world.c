/* world.c -- see world.h */
#include "world.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* _fioDetermineInterpCoord + fioGetPixelTrilinearInterp, R/src_common/FeatureIO.cpp:757-850 (host copy for
 * the one-off input resampling; the per-keypoint sampling runs on the GPU) */
static void interp_coord(float fX, float fMinX, float fMaxX, int *ix, float *w)
{
    if (fX < fMinX + 0.5f) {
        *ix = (int)fMinX;
        *w = 1.0f;
    } else if (fX >= fMaxX - 0.5f) {
        *ix = (int)(fMaxX - 2);
        *w = 0.0f;
    } else {
        float mh = fX - 0.5f;
        *ix = (int)floorf(mh);
        *w = 1.0f - (mh - ((float)*ix));
    }
}

static float trilinear(const float *img, int X, int Y, int Z, float x, float y, float z)
{
    float wx, wy, wz;
    int ix, iy, iz;
    interp_coord(x, 0, (float)X, &ix, &wx);
    interp_coord(y, 0, (float)Y, &iy, &wy);
    interp_coord(z, 0, (float)Z, &iz, &wz);
    const size_t XY = (size_t)X * Y;
    const float *p = img + (size_t)iz * XY + (size_t)iy * X + ix;
    float f000 = p[0], f100 = p[1], f010 = p[X], f110 = p[X + 1];
    float f001 = p[XY], f101 = p[XY + 1], f011 = p[XY + X], f111 = p[XY + X + 1];
    float fn00 = wx * f000 + (1.0f - wx) * f100;
    float fn01 = wx * f001 + (1.0f - wx) * f101;
    float fn10 = wx * f010 + (1.0f - wx) * f110;
    float fn11 = wx * f011 + (1.0f - wx) * f111;
    float fnn0 = wy * fn00 + (1.0f - wy) * fn10;
    float fnn1 = wy * fn01 + (1.0f - wy) * fn11;
    return wz * fnn0 + (1.0f - wz) * fnn1;
}

int sift3d_world_make_isotropic(nifti_min_image *img)
{
    if (!(img->dx != img->dy || img->dy != img->dz || img->dx != img->dz)) return 0;
    float fMin = img->dx;
    if (img->dy < fMin) fMin = img->dy;
    if (img->dz < fMin) fMin = img->dz;
    const int nx = (int)(img->nx * img->dx / fMin), ny = (int)(img->ny * img->dy / fMin), nz = (int)(img->nz * img->dz / fMin);
    if (nx <= 0 || ny <= 0 || nz <= 0) return -3;
    float *out = (float *)malloc(sizeof(float) * (size_t)nx * ny * nz);
    if (!out) return -3;
    float f[3] = {fMin / img->dx, fMin / img->dy, fMin / img->dz};
    /* one factor per matrix column: rescales the direction cosines (featExtract.cpp:162-171) */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            img->qto_xyz[i][j] *= f[j];
            if (img->sform_code > 0) img->sto_xyz[i][j] *= f[j];
        }
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++)
                out[((size_t)z * ny + y) * nx + x] =
                    trilinear(img->data, img->nx, img->ny, img->nz, (float)(x * f[0] + 0.5), (float)(y * f[1] + 0.5), (float)(z * f[2] + 0.5));
    free(img->data);
    img->data = out;
    img->nx = nx; img->ny = ny; img->nz = nz;
    img->dx = img->dy = img->dz = fMin;
    return 0;
}

/* invert_3x3<float,float>, R/src_common/MultiScale.h:192-222 */
static void invert3(float in[3][3], float out[3][3])
{
    float a11 = in[0][0], a21 = in[1][0], a31 = in[2][0];
    float a12 = in[0][1], a22 = in[1][1], a32 = in[2][1];
    float a13 = in[0][2], a23 = in[1][2], a33 = in[2][2];
    float det = a11 * (a33 * a22 - a32 * a23) - a21 * (a33 * a12 - a32 * a13) + a31 * (a23 * a12 - a22 * a13);
    float div = 1 / (float)det;
    out[0][0] = (a33 * a22 - a32 * a23) * div;
    out[1][0] = -(a33 * a21 - a31 * a23) * div;
    out[2][0] = (a32 * a21 - a31 * a22) * div;
    out[0][1] = -(a33 * a12 - a32 * a13) * div;
    out[1][1] = (a33 * a11 - a31 * a13) * div;
    out[2][1] = -(a32 * a11 - a31 * a12) * div;
    out[0][2] = (a23 * a12 - a22 * a13) * div;
    out[1][2] = -(a23 * a11 - a21 * a13) * div;
    out[2][2] = (a22 * a11 - a21 * a12) * div;
}

void sift3d_world_transform(sift3d_feature *recs, int64_t n, float m[4][4])
{
    float scale_sum = 0, rot[3][3];
    for (int i = 0; i < 3; i++) {
        /* vec3D_mag / vec3D_norm_3d of row i, R/src_common/MultiScale.cpp:1092-1127 */
        float ss = m[i][0] * m[i][0] + m[i][1] * m[i][1] + m[i][2] * m[i][2];
        scale_sum += ss > 0 ? sqrtf(ss) : 0;
        memcpy(rot[i], m[i], 3 * sizeof(float));
        if (ss > 0) {
            float div = (float)(1.0 / sqrtf(ss));
            rot[i][0] *= div; rot[i][1] *= div; rot[i][2] *= div;
        } else {
            rot[i][0] = 1; rot[i][1] = 0; rot[i][2] = 0;
        }
    }
    scale_sum /= 3;
    for (int64_t k = 0; k < n; k++) {
        sift3d_feature *r = &recs[k];
        float in[4] = {r->x, r->y, r->z, 1}, out[4];
        for (int i = 0; i < 4; i++) { /* mult_4x4_vector, MultiScale.cpp:1074-1089 */
            out[i] = 0;
            for (int j = 0; j < 4; j++) out[i] += m[i][j] * in[j];
        }
        r->x = out[0]; r->y = out[1]; r->z = out[2];
        r->scale *= scale_sum;
        float ori[3][3], a[3][3], b[3][3];
        memcpy(ori, r->ori, sizeof(ori));
        invert3(ori, a);
        for (int i = 0; i < 3; i++) /* mult_3x3_matrix<float,float>(pfRot3x3, pfOri, pfOriOut) */
            for (int j = 0; j < 3; j++) {
                b[i][j] = 0;
                for (int ii = 0; ii < 3; ii++) b[i][j] += rot[i][ii] * a[ii][j];
            }
        invert3(b, ori);
        memcpy(r->ori, ori, sizeof(ori));
    }
}
