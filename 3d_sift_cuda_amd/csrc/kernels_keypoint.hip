/*
 * kernels_keypoint.hip -- per-keypoint stage for gfx950: sub-voxel refinement,
 * 11^3 patch sampling, structure-tensor eigen test, canonical orientation
 * frames (phase A, one wavefront per extremum) and patch re-sampling +
 * 64-value descriptor + rank transform (phase B, one wavefront per output
 * record).  The patch and every histogram live in LDS.
 *
 * Parity contract: the reference does all of this in scalar CPU code whose
 * float accumulations are order dependent (R/src_common/MultiScale.cpp, R/ =
 * /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/).  Work is
 * spread over the 64 lanes only where the result does not depend on the
 * order (sampling, gradients, per-voxel terms, 3/5-tap patch blurs, peak
 * tests, rank counting); every order-dependent sum is kept as ONE sequential
 * chain in the reference's raster order, with one lane per independent chain
 * (9 lanes for the structure tensor, 8 lanes for the 8 corners of a splat, 64
 * lanes for the 64 descriptor bins).  Compiled with -ffp-contract=off.
 */
#include <cstddef>
#include <type_traits>

#include "sift3d_internal.h"

#define PD SIFT3D_PATCH_DIM
#define PV SIFT3D_PATCH_VOX

/* ---------------------------------------------------------------------- */
/* scalar helpers (each mirrors one reference routine)                     */
/* ---------------------------------------------------------------------- */

/* _fioDetermineInterpCoord, R/src_common/FeatureIO.cpp:757-782 */
__device__ __forceinline__ void interp_coord(float fX, float fMin, float fMax, int &ix, float &w)
{
    if (fX < fMin + 0.5f) {
        ix = (int)fMin;
        w = 1.0f;
    } else if (fX >= fMax - 0.5f) {
        ix = (int)(fMax - 2);
        w = 0.0f;
    } else {
        float mh = fX - 0.5f;
        ix = (int)floorf(mh);
        w = 1.0f - (mh - ((float)ix));
    }
}

/* fioGetPixelTrilinearInterp, R/src_common/FeatureIO.cpp:812-850 */
/* (x,y,z) are coordinates in the whole octave volume (Z = its slice count); img holds Zl slices
 * starting at global slice z_off, so cell and weights are those of the undivided volume and only
 * the address is shifted.  The clamp is a memory-safety net: with the 32-slice halo of Z-slab
 * mode (patch reach < 29 slices) it never binds. */
__device__ __forceinline__ float trilinear(const float *__restrict__ img, int X, int XP, int Y, int Z, int Zl, int z_off,
                                           float x, float y, float z)
{
    float wx, wy, wz;
    int ix, iy, iz;
    interp_coord(x, 0, (float)X, ix, wx);
    interp_coord(y, 0, (float)Y, iy, wy);
    interp_coord(z, 0, (float)Z, iz, wz);
    iz -= z_off;
    iz = iz < 0 ? 0 : (iz > Zl - 2 ? Zl - 2 : iz);
    const long long XY = (long long)XP * Y; /* XP: row pitch; X only bounds the coordinates */
    const float *p = img + (long long)iz * XY + (long long)iy * XP + ix;
    /* the two x-neighbours of a corner pair are adjacent in memory: four 8-byte gathers (dword aligned)
     * instead of eight 4-byte ones */
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    const f2u q00 = *reinterpret_cast<const f2u *>(p), q10 = *reinterpret_cast<const f2u *>(p + XP);
    const f2u q01 = *reinterpret_cast<const f2u *>(p + XY), q11 = *reinterpret_cast<const f2u *>(p + XY + XP);
    const float f000 = q00.x, f100 = q00.y, f010 = q10.x, f110 = q10.y;
    const float f001 = q01.x, f101 = q01.y, f011 = q11.x, f111 = q11.y;
    float fn00 = wx * f000 + (1.0f - wx) * f100;
    float fn01 = wx * f001 + (1.0f - wx) * f101;
    float fn10 = wx * f010 + (1.0f - wx) * f110;
    float fn11 = wx * f011 + (1.0f - wx) * f111;
    float fnn0 = wy * fn00 + (1.0f - wy) * fn10;
    float fnn1 = wy * fn01 + (1.0f - wy) * fn11;
    return wz * fnn0 + (1.0f - wz) * fnn1;
}

/* finddet + interpolate_extremum_quadratic, R/src_common/MultiScale.cpp:2531-2534, 1641-1697 */
__device__ __forceinline__ double finddet(double a1, double a2, double a3, double b1, double b2, double b3, double c1,
                                          double c2, double c3)
{
    return ((a1 * b2 * c3) - (a1 * b3 * c2) - (a2 * b1 * c3) + (a3 * b1 * c2) + (a2 * b3 * c1) - (a3 * b2 * c1));
}
__device__ __forceinline__ double interp_quadratic(double x0, double x1, double x2, double fx0, double fx1, double fx2)
{
    if (!(fx1 < fx0 && fx1 < fx2) && !(fx1 > fx0 && fx1 > fx2)) return x1;
    double a1 = x0 * x0, b1 = x0, c1 = 1;
    double a2 = x1 * x1, b2 = x1, c2 = 1;
    double a3 = x2 * x2, b3 = x2, c3 = 1;
    double d1 = fx0, d2 = fx1, d3 = fx2;
    double det = finddet(a1, a2, a3, b1, b2, b3, c1, c2, c3);
    double detx = finddet(d1, d2, d3, b1, b2, b3, c1, c2, c3);
    double dety = finddet(a1, a2, a3, d1, d2, d3, c1, c2, c3);
    if (d1 == 0 && d2 == 0 && d3 == 0) return x1;
    if (det != 0) {
        if (detx != 0) return dety / (-2.0 * detx);
    }
    return x1;
}

/* invert_3x3<float,double>, R/src_common/MultiScale.h:192-222 */
__device__ __forceinline__ void invert3(const float *in, float *out) /* row-major 3x3 */
{
    float a11 = in[0], a21 = in[3], a31 = in[6];
    float a12 = in[1], a22 = in[4], a32 = in[7];
    float a13 = in[2], a23 = in[5], a33 = in[8];
    float det = a11 * (a33 * a22 - a32 * a23) - a21 * (a33 * a12 - a32 * a13) + a31 * (a23 * a12 - a22 * a13);
    double div = 1 / (double)det;
    out[0] = (float)((a33 * a22 - a32 * a23) * div);
    out[3] = (float)(-(a33 * a21 - a31 * a23) * div);
    out[6] = (float)((a32 * a21 - a31 * a22) * div);
    out[1] = (float)(-(a33 * a12 - a32 * a13) * div);
    out[4] = (float)((a33 * a11 - a31 * a13) * div);
    out[7] = (float)(-(a32 * a11 - a31 * a12) * div);
    out[2] = (float)((a23 * a12 - a22 * a13) * div);
    out[5] = (float)(-(a23 * a11 - a21 * a13) * div);
    out[8] = (float)((a22 * a11 - a21 * a12) * div);
}

/* vec3D_norm_3d / vec3D_mag / vec3D_dot_3d, R/src_common/MultiScale.cpp:1092-1127, 1571-1578 */
__device__ __forceinline__ void v3_norm(float *p)
{
    float ss = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if (ss > 0) {
        float div = (float)(1.0 / (double)sqrtf(ss));
        p[0] *= div;
        p[1] *= div;
        p[2] *= div;
    } else {
        p[0] = 1;
        p[1] = 0;
        p[2] = 0;
    }
}
__device__ __forceinline__ float v3_mag(const float *p)
{
    float ss = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if (ss > 0) return sqrtf(ss);
    return 0;
}
__device__ __forceinline__ float v3_dot(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Singular value decomposition of the 3x3 structure tensor and the sort of its result: what the reference gets from
 * SingularValueDecomp<float,3,3> + SortEigenDecomp<float,3> (R/src_common/SVD.h:15-228, the Golub-Kahan-Reinsch scheme:
 * Householder reduction to bidiagonal form, accumulation of the right- and left-hand transformations, implicit-shift QR
 * sweeps).  The eigenvectors go into the records, so this has to produce the reference's bits, and for that every
 * floating-point operation has to be the reference's, in its order and in its width: storage in float, running values in
 * double, a product of two stored floats rounded to float before it enters a double sum.  Written here for the 3x3 case
 * with 0-based indices (the sizes folded in, the branches that cannot be taken removed) and pinned twice by the test
 * suite: its CPU restatement against the reference's own template compiled from source (bit-exact on 4 000 tensors), and
 * this function against that restatement through every record of the pipeline tests. */
__device__ __forceinline__ double svd_with_sign(double magnitude, double sign_source) { return sign_source >= 0.0 ? fabs(magnitude) : -fabs(magnitude); }
__device__ __forceinline__ double svd_hypot(double p, double q) { return sqrt(p * p + q * q); }

__device__ __forceinline__ void svd3(float a[3][3], float w[3], float v[3][3])
{
    double off[3]; /* the super-diagonal of the bidiagonal form (off[0] stays 0) */
    double g = 0.0, scale = 0.0, norm = 0.0, s, f, h;
    int below = 0;

    /* ---- 1. Householder reduction: column i (left reflection), then row i (right reflection) ---- */
    for (int i = 0; i < 3; i++) {
        below = i + 1;
        off[i] = scale * g;
        g = s = scale = 0.0;
        for (int k = i; k < 3; k++) scale += fabsf(a[k][i]);
        if (scale) {
            for (int k = i; k < 3; k++) {
                a[k][i] = (float)(a[k][i] / scale);
                s += a[k][i] * a[k][i];
            }
            f = a[i][i];
            g = -svd_with_sign(sqrt(s), f);
            h = f * g - s;
            a[i][i] = (float)(f - g);
            for (int j = below; j < 3; j++) {
                s = 0.0;
                for (int k = i; k < 3; k++) s += a[k][i] * a[k][j];
                f = s / h;
                for (int k = i; k < 3; k++) a[k][j] = (float)(a[k][j] + f * a[k][i]);
            }
            for (int k = i; k < 3; k++) a[k][i] = (float)(a[k][i] * scale);
        }
        w[i] = (float)(scale * g);
        g = s = scale = 0.0;
        if (i != 2) {
            for (int k = below; k < 3; k++) scale += fabsf(a[i][k]);
            if (scale) {
                for (int k = below; k < 3; k++) {
                    a[i][k] = (float)(a[i][k] / scale);
                    s += a[i][k] * a[i][k];
                }
                f = a[i][below];
                g = -svd_with_sign(sqrt(s), f);
                h = f * g - s;
                a[i][below] = (float)(f - g);
                for (int k = below; k < 3; k++) off[k] = a[i][k] / h;
                for (int j = below; j < 3; j++) {
                    s = 0.0;
                    for (int k = below; k < 3; k++) s += a[j][k] * a[i][k];
                    for (int k = below; k < 3; k++) a[j][k] = (float)(a[j][k] + s * off[k]);
                }
                for (int k = below; k < 3; k++) a[i][k] = (float)(a[i][k] * scale);
            }
        }
        const double row_norm = fabsf(w[i]) + fabs(off[i]);
        norm = norm > row_norm ? norm : row_norm;
    }

    /* ---- 2. right-hand transformations into v, last row first (g and `below` carry over from the row above) ---- */
    for (int i = 2; i >= 0; i--) {
        if (i < 2) {
            if (g) {
                for (int j = below; j < 3; j++) v[j][i] = (float)((a[i][j] / a[i][below]) / g);
                for (int j = below; j < 3; j++) {
                    s = 0.0;
                    for (int k = below; k < 3; k++) s += a[i][k] * v[k][j];
                    for (int k = below; k < 3; k++) v[k][j] = (float)(v[k][j] + s * v[k][i]);
                }
            }
            for (int j = below; j < 3; j++) v[i][j] = v[j][i] = 0.0;
        }
        v[i][i] = 1.0;
        g = off[i];
        below = i;
    }

    /* ---- 3. left-hand transformations, in place in a ---- */
    for (int i = 2; i >= 0; i--) {
        below = i + 1;
        g = w[i];
        for (int j = below; j < 3; j++) a[i][j] = 0.0;
        if (g) {
            g = 1.0 / g;
            for (int j = below; j < 3; j++) {
                s = 0.0;
                for (int k = below; k < 3; k++) s += a[k][i] * a[k][j];
                f = (s / a[i][i]) * g;
                for (int k = i; k < 3; k++) a[k][j] = (float)(a[k][j] + f * a[k][i]);
            }
            for (int j = i; j < 3; j++) a[j][i] = (float)(a[j][i] * g);
        } else {
            for (int j = i; j < 3; j++) a[j][i] = 0.0;
        }
        a[i][i] = a[i][i] + 1;
    }

    /* ---- 4. diagonalisation: for each singular value, from the last, implicit-shift QR sweeps (at most 30) ---- */
    for (int k = 2; k >= 0; k--) {
        for (int sweep = 1; sweep <= 30; sweep++) {
            /* find the block to work on: lo = first row of it; off[0] is 0, so the search always ends by splitting */
            bool cancel = true;
            int lo = k, above = 0;
            for (; lo >= 0; lo--) {
                above = lo - 1;
                if ((double)(fabs(off[lo]) + norm) == norm) {
                    cancel = false;
                    break;
                }
                if ((double)(fabsf(w[above]) + norm) == norm) break;
            }
            double c, x, y, z;
            if (cancel) { /* w[above] is negligible: rotate off[lo] .. off[k] away */
                c = 0.0;
                s = 1.0;
                for (int i = lo; i <= k; i++) {
                    f = s * off[i];
                    off[i] = c * off[i];
                    if ((double)(fabs(f) + norm) == norm) break;
                    g = w[i];
                    h = svd_hypot(f, g);
                    w[i] = (float)h;
                    h = 1.0 / h;
                    c = g * h;
                    s = -f * h;
                    for (int j = 0; j < 3; j++) {
                        y = a[j][above];
                        z = a[j][i];
                        a[j][above] = (float)(y * c + z * s);
                        a[j][i] = (float)(z * c - y * s);
                    }
                }
            }
            z = w[k];
            if (lo == k) { /* converged: make the singular value non-negative */
                if (z < 0.0) {
                    w[k] = (float)(-z);
                    for (int j = 0; j < 3; j++) v[j][k] = -v[j][k];
                }
                break;
            }
            /* shift from the bottom 2x2 minor */
            x = w[lo];
            above = k - 1;
            y = w[above];
            g = off[above];
            h = off[k];
            f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y);
            g = svd_hypot(f, 1.0);
            f = ((x - z) * (x + z) + h * ((y / (f + svd_with_sign(g, f))) - h)) / x;
            /* one QR transformation: a chase of Givens rotations down the block */
            c = s = 1.0;
            for (int j = lo; j <= above; j++) {
                const int i = j + 1;
                g = off[i];
                y = w[i];
                h = s * g;
                g = c * g;
                z = svd_hypot(f, h);
                off[j] = z;
                c = f / z;
                s = h / z;
                f = x * c + g * s;
                g = g * c - x * s;
                h = y * s;
                y *= c;
                for (int r = 0; r < 3; r++) {
                    x = v[r][j];
                    z = v[r][i];
                    v[r][j] = (float)(x * c + z * s);
                    v[r][i] = (float)(z * c - x * s);
                }
                z = svd_hypot(f, h);
                w[j] = (float)z;
                if (z) {
                    z = 1.0 / z;
                    c = f * z;
                    s = h * z;
                }
                f = c * g + s * y;
                x = c * y - s * g;
                for (int r = 0; r < 3; r++) {
                    y = a[r][j];
                    z = a[r][i];
                    a[r][j] = (float)(y * c + z * s);
                    a[r][i] = (float)(z * c - y * s);
                }
            }
            off[lo] = 0.0;
            off[k] = f;
            w[k] = (float)x;
        }
    }
}
__device__ __forceinline__ void sort_eig(float w[3], float v[3][3])
{
    float t;
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++)
            if (w[i] < w[j]) {
                t = w[j]; w[j] = w[i]; w[i] = t;
                for (int k = 0; k < 3; k++) {
                    t = v[k][j]; v[k][j] = v[k][i]; v[k][i] = t;
                }
            }
}
/* ---------------------------------------------------------------------- */
/* wave-cooperative building blocks (block = one wavefront of 64 lanes)    */
/* ---------------------------------------------------------------------- */
typedef float v4f __attribute__((ext_vector_type(4)));
/* One workgroup of KP_NT threads (4 wavefronts) per keypoint / record.  The LDS footprint is per
 * workgroup, so four wavefronts share what one used to hold alone: four times the wavefronts per CU for
 * the order-independent phases (gathers, gradients, pre-passes, patch blurs, peak tests), while the
 * sequential chains run on wavefront 0 and the others wait at the barrier. */
#define KP_NT 256   /* phase A: four wavefronts per keypoint */
#define DESC_NT 128  /* phase B: one wavefront per record (15 independent records per CU beat 8 four-wave workgroups) */
#define NRAD 515 /* voxels of the 11^3 patch with dx^2+dy^2+dz^2 < 25 (all of them interior) */
#define NRAD_PAD 516
#define NINT 729 /* interior voxels 1..9 in each axis */

/* sampleImage3D, R/src_common/MultiScale.cpp:2614-2714 (bounds test done by the caller) */
template <int NT>
__device__ __forceinline__ void wave_sample_patch(float *patch, const float *__restrict__ img, int X, int XP, int Y, int Z, int Zl,
                                                  int z_off, float fx, float fy, float fz, float scale, const float *ori9)
{
    float inv[9];
    invert3(ori9, inv);
    const float rad = 2.0f * scale;
    const int sr = PD / 2;
    const float sc = rad / (float)(sr);
    /* Which sample a lane takes changes no value, only which memory lines the 64 gathers of one instruction touch: the
     * lanes walk the patch axis that runs closest to the volume's x first (then the one closest to y), so that
     * neighbouring lanes read neighbouring voxels of a row wherever the frame allows it (descriptor kernel at 512^3:
     * 4.12 -> 3.97 ms; the patch in Z-curve blocks of 4 x 4 x 4 instead of lines: no better). */
    int ax_fast = 0, ax_mid = 1, ax_slow = 2;
    {
        const float f0 = fabsf(inv[0]), f1 = fabsf(inv[1]), f2 = fabsf(inv[2]);
        ax_fast = f0 >= f1 && f0 >= f2 ? 0 : (f1 >= f2 ? 1 : 2);
        const int r0 = ax_fast == 0 ? 1 : 0, r1 = ax_fast == 2 ? 1 : 2;
        const bool first = fabsf(inv[3 + r0]) >= fabsf(inv[3 + r1]);
        ax_mid = first ? r0 : r1;
        ax_slow = first ? r1 : r0;
    }
    const int st_fast = ax_fast == 0 ? 1 : (ax_fast == 1 ? PD : PD * PD), st_mid = ax_mid == 0 ? 1 : (ax_mid == 1 ? PD : PD * PD),
              st_slow = ax_slow == 0 ? 1 : (ax_slow == 1 ? PD : PD * PD);
    /* three samples per lane per trip: the 24 gathers of a trip are issued together */
    for (int s0 = threadIdx.x; s0 < PV; s0 += 3 * NT) {
        float pix[3];
        int sidx[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int t = s0 + NT * u;
            pix[u] = 0;
            sidx[u] = 0;
            if (t < PV) {
                const int s = (t % PD) * st_fast + ((t / PD) % PD) * st_mid + (t / (PD * PD)) * st_slow; /* the patch voxel this lane samples */
                sidx[u] = s;
                const int xx = s % PD - sr, yy = (s / PD) % PD - sr, zz = s / (PD * PD) - sr;
                float in3[3] = {(float)xx, (float)yy, (float)zz};
                float o[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    float a = 0;
#pragma unroll
                    for (int j = 0; j < 3; j++) a += inv[i * 3 + j] * in3[j];
                    o[i] = a;
                }
                o[0] *= sc; o[1] *= sc; o[2] *= sc;
                o[0] += fx; o[1] += fy; o[2] += fz;
                if (o[0] < 0 || o[0] >= X) pix[u] = 0;
                else pix[u] = trilinear(img, X, XP, Y, Z, Zl, z_off, o[0], o[1], o[2]);
            }
        }
#pragma unroll
        for (int u = 0; u < 3; u++)
            if (s0 + NT * u < PV) patch[sidx[u]] = pix[u];
    }
    __syncthreads();
}

/* One sequential float sum over the 1331 patch values (optionally of their
 * squares), in raster order, by one lane; 16-byte LDS reads. d is 16-byte aligned. */
template <bool SQUARE>
__device__ __forceinline__ float serial_sum_patch(const float *d)
{
    /* 1331 = 41 blocks of 32 + 19: read a block of 32 values (eight 16-byte LDS reads) while the
     * previous block is being added, so the LDS latency stays off the add chain */
    float acc = 0;
    const v4f *d4 = reinterpret_cast<const v4f *>(d);
    /* two register buffers used alternately (the block loop unrolled by two): a single buffer pair with a copy at the end
     * of every block cost 32 register moves per 32 additions, doubling the instructions of a chain that runs on one lane */
    v4f a[8], b[8];
    auto add_block = [&](const v4f(&blk)[8]) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const v4f v = blk[q];
            if (SQUARE) {
                acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
            } else {
                acc += v.x; acc += v.y; acc += v.z; acc += v.w;
            }
        }
    };
#pragma unroll
    for (int q = 0; q < 8; q++) a[q] = d4[q];
    for (int blk = 0; blk + 1 < 41; blk += 2) { /* blocks blk (in a) and blk + 1 (into b), then blk + 2 into a */
#pragma unroll
        for (int q = 0; q < 8; q++) b[q] = d4[(blk + 1) * 8 + q];
        add_block(a);
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = d4[(blk + 2) * 8 + q]; /* blk + 2 <= 40: the last block, added after the loop */
        add_block(b);
    }
    add_block(a); /* block 40 */
    for (int i = 41 * 32; i < PV; i++) acc += SQUARE ? d[i] * d[i] : d[i];
    return acc;
}

/* Feature3D::NormalizeData, R/src_common/MultiScale.cpp:127-205: the two
 * 1331-term sums are single sequential chains (lane 0). */
template <int NT>
__device__ __forceinline__ void wave_normalize_patch(float *d, float *scratch2)
{
    if (threadIdx.x == 0) scratch2[0] = serial_sum_patch<false>(d) / (PD * PD * PD);
    __syncthreads();
    const float mean = scratch2[0];
    for (int i = threadIdx.x; i < PV; i += NT) d[i] -= mean;
    __syncthreads();
    if (threadIdx.x == 0) scratch2[1] = 1.0f / sqrtf(serial_sum_patch<true>(d));
    __syncthreads();
    const float div = scratch2[1];
    for (int i = threadIdx.x; i < PV; i += NT) d[i] *= div;
    __syncthreads();
}

__device__ __forceinline__ bool in_radius(int s)
{
    const int x = s % PD - PD / 2, y = (s / PD) % PD - PD / 2, z = s / (PD * PD) - PD / 2;
    const float fx = (float)x, fy = (float)y, fz = (float)z;
    return fz * fz + fy * fy + fx * fx < (float)((PD / 2) * (PD / 2));
}

/* Raster-ordered list of the in-radius voxels (ballot compaction by wavefront 0); returns the count (515). */
__device__ __forceinline__ int wave_build_radius_list(unsigned short *rlist, int *n_lds)
{
    if (threadIdx.x < 64) {
        int n = 0;
        for (int base = 0; base < PV; base += 64) {
            const int s = base + threadIdx.x;
            const bool in = s < PV && in_radius(s);
            const unsigned long long m = __ballot(in);
            if (in) rlist[n + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = (unsigned short)s;
            n += __popcll(m);
        }
        if (threadIdx.x == 0) *n_lds = n;
    }
    __syncthreads();
    return *n_lds;
}

/* One pass of the separable patch blur: a thread takes a whole line of the axis and slides the filter window along it,
 * one LDS read per output instead of one per tap (the per-keypoint kernels are bound by the LDS unit).  Per output the
 * same chain as before: acc = 0, then taps in ascending order, positions outside the patch skipped. */
template <int NT>
__device__ __forceinline__ void blur_pass(const float *src, float *dst, int axis, const float *taps, int ntaps)
{
    const int h = ntaps / 2;
    const int st = axis == 0 ? 1 : (axis == 1 ? PD : PD * PD);
    float tp[5], w[5];
#pragma unroll
    for (int j = 0; j < 5; j++) tp[j] = j < ntaps ? taps[j] : 0.0f;
    for (int ln = threadIdx.x; ln < PD * PD; ln += NT) {
        const int base = axis == 0 ? ln * PD : (axis == 1 ? (ln / PD) * PD * PD + ln % PD : ln);
#pragma unroll
        for (int j = 0; j < 5; j++) { /* window of output 0: positions -h .. +h */
            const int pos = j - h;
            w[j] = (j < ntaps && pos >= 0 && pos < PD) ? src[base + pos * st] : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < PD; c++) {
            float acc = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int cc = c + j - h;
                if (j < ntaps && cc >= 0 && cc < PD) acc = acc + tp[j] * w[j];
            }
            dst[base + c * st] = acc;
            const int nx = c + 1 + h; /* the position entering the window */
            const float in = nx < PD ? src[base + nx * st] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = w[j + 1];
            if (ntaps >= 1) w[ntaps - 1 < 4 ? ntaps - 1 : 4] = in;
        }
    }
    __syncthreads();
}
/* 3- or 5-tap separable blur of an 11^3 LDS volume, pass order x,y,z, zero
 * borders (gb3d_blur3d on the patch: R/src_common/MultiScale.cpp:2850,2972,1032).
 * The result lands in tmp_a (in -> tmp_a -> tmp_b -> tmp_a; tmp_b may be the input buffer); taps are in LDS. */
template <int NT>
__device__ __forceinline__ void wave_blur_patch(float *in, float *tmp_a, float *tmp_b, const float *taps, int ntaps)
{
    blur_pass<NT>(in, tmp_a, 0, taps, ntaps);
    blur_pass<NT>(tmp_a, tmp_b, 1, taps, ntaps);
    blur_pass<NT>(tmp_b, tmp_a, 2, taps, ntaps);
}

/* regFindFEATUREIOPeaks without callback (R/src_common/MultiScale.cpp:1987-2121)
 * + lvSortHighLow (R/src_common/LocationValue.cpp:28-56, stable): every thread tests its voxels
 * and leaves a flag, wavefront 0 compacts the flags in raster order (ballot), then a stable
 * descending rank by counting.  Returns the count; pk_idx/pk_val hold the sorted list.
 * flags: PV bytes of scratch LDS; n_lds: one LDS int. */
template <int NT>
__device__ __forceinline__ int wave_peaks_sorted(const float *g, unsigned char *flags, int *n_lds, short *raw_idx, float *raw_val,
                                                 short *pk_idx, float *pk_val)
{
    /* One thread per interior row (y, z in 1..9) slides a 3-column window of the nine neighbouring rows along x: nine
     * LDS reads per voxel instead of twenty-seven (the kernel is bound by its LDS unit); the comparisons are the same. */
    static_assert(NT >= 81 + PD * PD, "one thread per interior row plus one per row for the border flags");
    const int t = threadIdx.x;
    if (t < 81) {
        const int y = t % 9 + 1, z = t / 9 + 1;
        const int rb = (z * PD + y) * PD; /* the row's first voxel */
        float a[9], b[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int off = rb + ((k / 3 - 1) * PD + (k % 3 - 1)) * PD;
            a[k] = g[off];
            b[k] = g[off + 1];
        }
        flags[rb] = 0;
#pragma unroll
        for (int x = 1; x < PD - 1; x++) {
            float n[9];
#pragma unroll
            for (int k = 0; k < 9; k++) n[k] = g[rb + ((k / 3 - 1) * PD + (k % 3 - 1)) * PD + x + 1];
            const float c = b[4];
            bool pk = true;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                pk = pk && (a[k] < c) && (n[k] < c);
                if (k != 4) pk = pk && (b[k] < c);
            }
            flags[rb + x] = pk ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                a[k] = b[k];
                b[k] = n[k];
            }
        }
        flags[rb + PD - 1] = 0;
    } else if (t < 81 + PD * PD) { /* rows on a face of the patch hold no peaks */
        const int q = t - 81, y = q % PD, z = q / PD;
        if (y == 0 || y == PD - 1 || z == 0 || z == PD - 1) {
#pragma unroll
            for (int x = 0; x < PD; x++) flags[(z * PD + y) * PD + x] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        int n = 0;
        for (int base = 0; base < PV; base += 64) {
            const int s = base + threadIdx.x;
            const bool pk = s < PV && flags[s] != 0;
            const unsigned long long m = __ballot(pk);
            if (pk) {
                const int pos = n + __popcll(m & ((1ull << threadIdx.x) - 1ull));
                raw_idx[pos] = (short)s;
                raw_val[pos] = g[s];
            }
            n += __popcll(m);
        }
        if (threadIdx.x == 0) *n_lds = n;
    }
    __syncthreads();
    const int n = *n_lds;
    for (int i = threadIdx.x; i < n; i += NT) {
        const float vi = raw_val[i];
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const float vj = raw_val[j];
            rank += (vj > vi || (vj == vi && j < i)) ? 1 : 0;
        }
        pk_idx[rank] = raw_idx[i];
        pk_val[rank] = vi;
    }
    __syncthreads();
    return n;
}

/* interpolate_discrete_3D_point on an 11^3 grid, R/src_common/MultiScale.cpp:1614-1639 */
__device__ __forceinline__ void interp_point_patch(const float *g, int s, float *o)
{
    const int ix = s % PD, iy = (s / PD) % PD, iz = s / (PD * PD);
    const float c = g[s];
    o[0] = (float)interp_quadratic(ix - 1, ix, ix + 1, g[s - 1], c, g[s + 1]);
    o[1] = (float)interp_quadratic(iy - 1, iy, iy + 1, g[s - PD], c, g[s + PD]);
    o[2] = (float)interp_quadratic(iz - 1, iz, iz + 1, g[s - PD * PD], c, g[s + PD * PD]);
}

/* Splat parameters of one in-radius voxel: the cell of corner 000, packed ix | iy<<4 | iz<<8, and
 * the three weights of _fioDetermineInterpCoord for the position (x,y,z) on the 11^3 grid. */
__device__ __forceinline__ void splat_params(float x, float y, float z, short &base, float &wx, float &wy, float &wz)
{
    int ix, iy, iz;
    interp_coord(x, 0, (float)PD, ix, wx);
    interp_coord(y, 0, (float)PD, iy, wy);
    interp_coord(z, 0, (float)PD, iz, wz);
    base = (short)(ix | (iy << 4) | (iz << 8));
}

/* Sequential trilinear splat of the in-radius voxels into an 11^3 grid
 * (fioIncPixelTrilinearInterp, R/src_common/FeatureIO.cpp:853-889), voxels in
 * raster order.  64 lanes = 8 consecutive voxels x 8 corners: cell index and
 * contribution value*wx'*wy'*wz' are computed for eight voxels at once (order
 * independent), then the eight voxels are applied one after the other, each
 * as one 8-lane read-modify-write of eight different cells.  LDS operations
 * of a wavefront execute in issue order, so every cell sees its updates in
 * voxel order, which is the reference's.  A voxel the reference skips has
 * mag == 0 and adds +0, which leaves a cell unchanged. */
__device__ __forceinline__ void wave_splat_sequence(float *grid, int n, const short *sp_base, const v4f *sp4)
{
    const int lane = threadIdx.x;
    const int slot = lane >> 3;
    const int a = lane & 1, b = (lane >> 1) & 1, c = (lane >> 2) & 1;
    float *g = grid; /* plain LDS accesses: a volatile generic pointer would turn them into system-scope flat ops */
    if (lane < 64) { /* wavefront 0 carries the chain */
        /* parameters of the group after the current one are read while the current one is being applied */
        int packed = 0;
        float wx = 0, wy = 0, wz = 0, v = 0;
        if (slot < n) { /* (wx, wy, wz, magnitude) of a voxel in one 16-byte read */
            packed = sp_base[slot];
            const v4f q = sp4[slot];
            wx = q.x; wy = q.y; wz = q.z; v = q.w;
        }
        for (int g0 = 0; g0 < n; g0 += 8) {
            const int i = g0 + slot;
            const bool ok = i < n;
            const int ix = packed & 15, iy = (packed >> 4) & 15, iz = packed >> 8;
            const int cell = ((iz + c) * PD + (iy + b)) * PD + (ix + a);
            const float ux = a ? (1.0f - wx) : wx;
            const float uy = b ? (1.0f - wy) : wy;
            const float uz = c ? (1.0f - wz) : wz;
            const float contrib = v * ux * uy * uz;
            const int in = i + 8;
            if (in < n) {
                packed = sp_base[in];
                const v4f q = sp4[in];
                wx = q.x; wy = q.y; wz = q.z; v = q.w;
            }
            /* ds_add_f32 without return: the LDS unit performs the IEEE single-precision add (round to nearest
             * even, denormals kept: the same operation as v_add_f32, checked by sift3d_selftest_lds_add) in the
             * order the instructions were issued, and the wavefront does not wait for any of them */
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (slot == j && ok) __hip_atomic_fetch_add(&g[cell], contrib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
}

/* Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one, with its private L2), and the
 * work items arrive sorted by (level, raster index), so neighbours in the list sample neighbouring
 * image regions: an XCD takes RUNS of consecutive items instead of every eighth item, and the trilinear
 * gathers of neighbouring keypoints / of the frames of one keypoint meet in one L2.
 * Round 4: the runs are short -- the list is cut into segments of seg items (256 by default) and each
 * segment is dealt to the XCDs in contiguous eighths -- so that all eight XCDs are in the same part of the
 * list at the same time.  With one eighth of the WHOLE list per XCD (rounds 2 - 3) the eighths took
 * different times (the records of the first detection levels sample compact footprints, the later ones
 * sparse ones) and the kernel ended with some XCDs idle for a tenth of its time: 9.60 -> 9.23 ms per 512^3
 * extraction (profiles/r04_desc_segment.txt).  Placement only changes speed, never results. */
__device__ __forceinline__ long long xcd_contiguous_item(long long n, int seg)
{
    const long long b = blockIdx.x;
    if (seg <= 0) {
        const long long per = (n + 7) / 8;
        return (b % 8) * per + b / 8;
    }
    /* the list in segments of seg items (a multiple of 8), each dealt to the XCDs in contiguous eighths: every XCD still walks
     * neighbouring items, and all eight are in the same part of the list at the same time */
    const long long s0 = b / seg * seg, local = b - s0;
    return s0 + (local % 8) * (seg / 8) + local / 8;
}

/* ---------------------------------------------------------------------- */
/* Phase A: extremum -> keypoint (geometry, eigen test, orientation frames) */
/* ---------------------------------------------------------------------- */
/* 23 032 bytes: seven workgroups per CU (the kernel is latency-bound; its throughput follows the number of resident
 * workgroups).  Buffers are shared between phases that never overlap:
 *   A   patch -> blurred grid                            B   splat grid t0 / middle pass of the blur
 *   A+C (contiguous) splat parameters (wx, wy, wz, magnitude) of the in-radius voxels, one 16-byte vector each
 *   C   scratch of the peak search                       r   in-radius list -> splat base cells */
struct kpA_smem {
    float A[PV + 1];
    union { /* directly behind A */
        float sp_tail[2 * NRAD_PAD]; /* the splat parameters run from A into here: live from the pre-pass to the end of the splat */
        struct {                    /* live from the peak search to the end of the frame loop body */
            unsigned char flags[PV + 13];
            short raw_idx[128], pk2_idx[128];
            float raw_val[128], pk2_val[128];
        } pk;
    } C;
    float B[PV + 1];
    float gx[NRAD_PAD], gy[NRAD_PAD], gz[NRAD_PAD]; /* gradients of the in-radius voxels */
    union {
        unsigned short rlist[NRAD_PAD]; /* until the gradients are taken */
        short sp_base[NRAD_PAD];        /* from the first splat pre-pass on */
    } r;
    short pk_idx[128];
    float pk_val[128];
    float ori_data[PD * 3 + 3];
    float sc[16];
    float taps[8];
    int cnt[4];
};
static_assert(sizeof(kpA_smem) * 7 <= 160 * 1024, "seven workgroups per CU");
static_assert(offsetof(kpA_smem, C) == sizeof(float) * (PV + 1) && sizeof(float) * (PV + 1) + sizeof(float) * 2 * NRAD_PAD >= 16 * NRAD_PAD,
              "the splat parameters span A and C");

__global__ __launch_bounds__(KP_NT, 7) void keypoint_kernel(sift3d_kp_params p, const unsigned long long *__restrict__ keys,
                                                      const sift3d_cval *__restrict__ vals, long long ncand,
                                                      sift3d_dkp *__restrict__ kps, int *__restrict__ nrec_out,
                                                      sift3d_taps taps3)
{
    __shared__ __attribute__((aligned(16))) kpA_smem sm;
    const long long k = blockIdx.x;
    if (k >= ncand) return;
    const int lane = threadIdx.x;
    if (lane < 3) sm.taps[lane] = taps3.f[lane];
    const unsigned long long key = keys[k];
    const sift3d_cval cvl = vals[k];
    const int lvl = (int)(key >> SIFT3D_KEY_LVL_SHIFT);
    const int is_max = (int)((key >> SIFT3D_KEY_MAX_SHIFT) & 1ull);
    const long long cidx = (long long)(key & SIFT3D_KEY_IDX_MASK);
    const sift3d_level lv = p.levels[lvl];
    sift3d_dkp *kp = kps + k;
    const int X = lv.X, Y = lv.Y, Z = lv.Z, XP = lv.XP;
    const long long XY = (long long)XP * Y;
    /* cidx indexes the local buffer; refinement and all geometry use the global slice number */
    const int ix = (int)(cidx % XP), iy = (int)((cidx / XP) % Y), iz = (int)(cidx / XY) + lv.z_off;

    /* generateFeatures3D_efficient, R/src_common/MultiScale.cpp:1361-1421 (every lane, identical) */
    const float *C = lv.dogc;
    const float cv = C[cidx];
    float fx = (float)interp_quadratic(ix - 1, ix, ix + 1, C[cidx - 1], cv, C[cidx + 1]);
    float fy = (float)interp_quadratic(iy - 1, iy, iy + 1, C[cidx - XP], cv, C[cidx + XP]);
    float fz = (float)interp_quadratic(iz - 1, iz, iz + 1, C[cidx - XY], cv, C[cidx + XY]);
    float scale = (float)(2 * interp_quadratic(lv.sigma_h, lv.sigma_c, lv.sigma_l, cvl.h, cv, cvl.l));
    fx += 0.5f; fy += 0.5f; fz += 0.5f;

    /* sampleImage3D bounds test, MultiScale.cpp:2630-2643 */
    const float rad = 2.0f * scale;
    const int rmax = (int)(rad + 2);
    if (fx - rmax < 0 || fy - rmax < 0 || fz - rmax < 0 || fx + rmax >= X || fy + rmax >= Y || fz + rmax >= Z) {
        if (lane == 0) nrec_out[k] = 0;
        return;
    }
    float *patch = sm.A;
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    wave_sample_patch<KP_NT>(patch, lv.img, X, lv.XP, Y, Z, lv.Zl, lv.z_off, fx, fy, fz, scale, ident);
#ifdef SIFT3D_DEV
    if (p.debug_stop == 1) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
    wave_normalize_patch<KP_NT>(patch, sm.sc);
#ifdef SIFT3D_DEV
    if (p.debug_stop == 2) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
    const int nrad = wave_build_radius_list(sm.r.rlist, &sm.cnt[0]);

    /* determineOrientation3D, MultiScale.cpp:2541-2607: gradients (fioGenerateEdgeImages3D,
     * FeatureIO.cpp:2284-2326) are only ever used inside the radius */
    for (int i = lane; i < NRAD_PAD; i += KP_NT) {
        float a = 0, b = 0, c = 0;
        if (i < nrad) {
            const int s = sm.r.rlist[i];
            a = patch[s + 1] - patch[s - 1];
            b = patch[s + PD] - patch[s - PD];
            c = patch[s + PD * PD] - patch[s - PD * PD];
        }
        sm.gx[i] = a; sm.gy[i] = b; sm.gz[i] = c;
    }
    __syncthreads();
    if (lane < 9) {
        const float *ei = lane / 3 == 0 ? sm.gx : (lane / 3 == 1 ? sm.gy : sm.gz);
        const float *ej = lane % 3 == 0 ? sm.gx : (lane % 3 == 1 ? sm.gy : sm.gz);
        /* one sequential chain per matrix entry; the LDS reads of a block of 16 terms are issued together so
         * that their latency stays off the add chain */
        float acc = 0;
        int i = 0;
        for (; i + 16 <= nrad; i += 16) { /* 16-byte LDS reads: the kernel is bound by its LDS unit */
            v4f a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                a[q] = *reinterpret_cast<const v4f *>(ei + i + 4 * q);
                b[q] = *reinterpret_cast<const v4f *>(ej + i + 4 * q);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                acc += a[q].x * b[q].x; acc += a[q].y * b[q].y; acc += a[q].z * b[q].z; acc += a[q].w * b[q].w;
            }
        }
        for (; i < nrad; i++) acc += ei[i] * ej[i];
        sm.sc[lane] = acc;
    }
    __syncthreads();
#ifdef SIFT3D_DEV
    if (p.debug_stop == 3) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
    /* The SVD + eigen test (one lane, double-precision internals) and the first orientation-histogram splat (one
     * wavefront, a chain of LDS atomics) do not depend on each other: the splat runs on wavefront 0 while lane 0 of
     * the last wavefront does the SVD.  For the ~15 % of the extrema the eigen test rejects the splat was wasted. */
    auto svd_and_eigen_test = [&]() {
        float mat[3][3], w[3], v[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                mat[i][j] = sm.sc[i * 3 + j];
                v[i][j] = 0;
            }
        w[0] = w[1] = w[2] = 0;
        svd3(mat, w, v);
        sort_eig(w, v);
        /* eigen test, MultiScale.cpp:1748-1769 */
        float es = w[0] + w[1] + w[2];
        float ep = w[0] * w[1] * w[2];
        float esp = es * es * es;
        int keep = (esp < p.eig_thres * ep || p.eig_thres < 0) ? 1 : 0;
        sm.sc[15] = keep ? 1.0f : 0.0f;
        if (keep) {
            kp->x = fx; kp->y = fy; kp->z = fz; kp->scale = scale;
            kp->eigs[0] = w[0]; kp->eigs[1] = w[1]; kp->eigs[2] = w[2];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) kp->ori0[i * 3 + j] = v[i][j];
            kp->info = is_max ? SIFT3D_INFO_MIN0MAX1 : 0u;
            kp->lvl = lvl;
        } else {
            nrec_out[k] = 0;
        }
    };
#ifdef SIFT3D_DEV
    /* development build, stop code 40 (round-4 review item 5b): the SVD and the eigen test BEFORE the first splat instead of
     * beside it -- the 16 % of the extrema the test rejects then skip the patch hand-over, the splat pre-pass and the splat;
     * the others lose the overlap of the SVD (one lane, ~0.7 us of double-precision chain) with the splat */
    const bool svd_first = p.debug_stop == 40;
    if (svd_first) {
        if (lane == KP_NT - 64) svd_and_eigen_test();
        __syncthreads();
        if (sm.sc[15] == 0.0f) return;
    }
#else
    constexpr bool svd_first = false;
#endif
    /* the un-reoriented record is described from exactly this patch (identity frame, normalised once): hand it to
     * phase B instead of having it gathered from the image a second time (unused if the eigen test rejects) */
    for (int s = lane; s < PV; s += KP_NT) p.patch0[(long long)k * PV + s] = patch[s];
    __syncthreads(); /* the splat parameters below overwrite the patch */

    /* determineCanonicalOrientation3D, MultiScale.cpp:2722-3037.  The patch (A) is dead from here on. */
    float *t0 = sm.B;
    float *ta = sm.A;                        /* blur: t0 -> ta -> t0 -> ta */
    v4f *sp4 = reinterpret_cast<v4f *>(sm.A); /* live only between the pre-pass and the end of the splat */
    const float radius = (float)(PD / 2);
    for (int s = lane; s < PV; s += KP_NT) t0[s] = 0;
    for (int i = lane; i < nrad; i += KP_NT) {
        float e[3] = {sm.gx[i], sm.gy[i], sm.gz[i]};
        float m2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        float mg = 0, wx = 0, wy = 0, wz = 0;
        short base = 0;
        if (m2 != 0) {
            mg = sqrtf(m2);
            float u[3];
            for (int q = 0; q < 3; q++) u[q] = e[q] * radius / mg;
            for (int q = 0; q < 3; q++) u[q] += radius;
            splat_params((float)(u[0] + 0.5), (float)(u[1] + 0.5), (float)(u[2] + 0.5), base, wx, wy, wz);
        }
        sm.r.sp_base[i] = base;
        v4f q; q.x = wx; q.y = wy; q.z = wz; q.w = mg;
        sp4[i] = q;
    }
    __syncthreads();
    if (!svd_first && lane == KP_NT - 64) svd_and_eigen_test();
    wave_splat_sequence(t0, nrad, sm.r.sp_base, sp4); /* wavefront 0; ends with a barrier */
    if (sm.sc[15] == 0.0f) return;
#ifdef SIFT3D_DEV
    if (p.debug_stop == 4 || p.debug_stop == 5) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
#ifdef SIFT3D_DEV
    if (p.debug_stop == 6) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
    wave_blur_patch<KP_NT>(t0, ta, t0, sm.taps, 3);
#ifdef SIFT3D_DEV
    if (p.debug_stop == 7) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
    const int npk = wave_peaks_sorted<KP_NT>(ta, sm.C.pk.flags, &sm.cnt[1], sm.C.pk.raw_idx, sm.C.pk.raw_val, sm.pk_idx, sm.pk_val);
#ifdef SIFT3D_DEV
    if (p.debug_stop == 8) { if (lane == 0) nrec_out[k] = 0; return; }
#endif

    if (lane < npk && lane < PD && lane < 30) {
        float o[3];
        interp_point_patch(ta, sm.pk_idx[lane], o);
        o[0] -= radius; o[1] -= radius; o[2] -= radius;
        v3_norm(o);
        sm.ori_data[lane * 3 + 0] = o[0];
        sm.ori_data[lane * 3 + 1] = o[1];
        sm.ori_data[lane * 3 + 2] = o[2];
    }
    __syncthreads();

    int nret = 0;
    const float pk0 = npk > 0 ? sm.pk_val[0] : 0.0f;
    for (int i = 0; i < npk && i < PD && nret < 30; i++) {
        if ((double)sm.pk_val[i] < 0.8 * (double)pk0) break;
        const float p1[3] = {sm.ori_data[i * 3], sm.ori_data[i * 3 + 1], sm.ori_data[i * 3 + 2]};
        __syncthreads();
        for (int s = lane; s < PV; s += KP_NT) t0[s] = 0;
        for (int q = lane; q < nrad; q += KP_NT) {
            float e[3] = {sm.gx[q], sm.gy[q], sm.gz[q]};
            float mg = v3_mag(e);
            float wx = 0, wy = 0, wz = 0;
            short base = 0;
            if (mg != 0) {
                float u[3] = {e[0], e[1], e[2]};
                v3_norm(u);
                float par = v3_dot(p1, u);
                float pp[3];
                pp[0] = u[0] - par * p1[0];
                pp[1] = u[1] - par * p1[1];
                pp[2] = u[2] - par * p1[2];
                v3_norm(pp);
                for (int r = 0; r < 3; r++) {
                    pp[r] *= radius;
                    pp[r] += radius;
                }
                splat_params((float)(pp[0] + 0.5), (float)(pp[1] + 0.5), (float)(pp[2] + 0.5), base, wx, wy, wz);
            }
            sm.r.sp_base[q] = base;
            v4f pq; pq.x = wx; pq.y = wy; pq.z = wz; pq.w = mg;
            sp4[q] = pq;
        }
        __syncthreads();
#ifdef SIFT3D_DEV
        if (p.debug_stop == 31) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
        wave_splat_sequence(t0, nrad, sm.r.sp_base, sp4);
#ifdef SIFT3D_DEV
        if (p.debug_stop == 32) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
        wave_blur_patch<KP_NT>(t0, ta, t0, sm.taps, 3);
#ifdef SIFT3D_DEV
        if (p.debug_stop == 33) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
        const int npk2 = wave_peaks_sorted<KP_NT>(ta, sm.C.pk.flags, &sm.cnt[2], sm.C.pk.raw_idx, sm.C.pk.raw_val, sm.C.pk.pk2_idx, sm.C.pk.pk2_val);
#ifdef SIFT3D_DEV
        if (p.debug_stop == 34) { if (lane == 0) nrec_out[k] = 0; return; }
#endif
        const float pk20 = npk2 > 0 ? sm.C.pk.pk2_val[0] : 0.0f;
        for (int j = 0; j < npk2 && nret < PD && nret < 30; j++) {
            if (sm.C.pk.pk2_val[j] < 0.5f * pk20) break;
            if (lane == 0) {
                float p2[3], p3[3];
                interp_point_patch(ta, sm.C.pk.pk2_idx[j], p2);
                p2[0] -= radius; p2[1] -= radius; p2[2] -= radius;
                v3_norm(p2);
                float par = v3_dot(p1, p2);
                p2[0] = p2[0] - par * p1[0];
                p2[1] = p2[1] - par * p1[1];
                p2[2] = p2[2] - par * p1[2];
                v3_norm(p2);
                p3[0] = p1[1] * p2[2] - p1[2] * p2[1];
                p3[1] = -p1[0] * p2[2] + p1[2] * p2[0];
                p3[2] = p1[0] * p2[1] - p1[1] * p2[0];
                float *m = kp->frames + 9 * nret;
                for (int iv = 0; iv < 3; iv++) {
                    m[0 * 3 + iv] = p1[iv];
                    m[1 * 3 + iv] = p2[iv];
                    m[2 * 3 + iv] = p3[iv];
                }
            }
            nret++;
        }
    }
    if (lane == 0) {
        kp->nframes = nret;
        nrec_out[k] = 1 + nret;
    }
}

/* ---------------------------------------------------------------------- */
/* Phase B: one output record per wavefront: re-sample, normalise, describe */
/* ---------------------------------------------------------------------- */
/* value of lane i (compile-time i) of a wavefront-wide float: v_readlane_b32 */
__device__ __forceinline__ float lane_value(float x, int i)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), i));
}

/* pair tables of msGenerateBRIEFindex method 2 (data, R/src_common/MultiScale.cpp:805-807) */
__constant__ unsigned char c_brief_x[192] = {
    5,4,4,4,4,2,6,5,5,4,4,4,3,8,5,5,6,3,5,5,5,5,6,5,4,6,6,6,3,4,4,4,5,3,4,5,4,5,5,4,2,7,7,5,3,5,4,5,3,5,7,3,5,5,2,3,5,5,6,6,4,6,5,4,
    4,6,5,3,5,6,4,3,6,4,4,5,3,3,3,6,6,5,2,4,4,6,3,6,3,2,3,5,4,5,3,4,3,6,5,4,3,6,4,5,2,4,3,7,2,3,6,5,2,6,3,3,5,6,3,6,3,5,3,6,5,7,4,2,
    5,5,5,2,5,7,4,2,5,3,4,3,3,7,4,4,7,6,4,4,2,8,7,6,5,4,7,3,6,6,5,2,4,5,3,2,5,5,1,6,3,6,3,6,2,5,4,4,7,2,6,3,2,2,4,3,3,2,3,4,2,5,6,7};
__constant__ unsigned char c_brief_y[192] = {
    6,5,3,4,5,3,7,4,6,4,3,2,4,7,5,3,5,1,5,4,7,6,8,4,4,5,6,5,2,5,4,6,4,0,4,3,3,4,4,2,1,7,8,6,4,4,1,6,1,3,7,2,3,3,1,3,6,1,6,6,4,7,6,4,
    3,5,4,2,3,6,4,5,6,3,3,5,1,3,1,6,7,4,1,4,3,5,2,4,2,1,2,5,4,5,2,3,3,3,3,4,2,6,3,4,3,3,3,6,1,2,5,4,2,4,1,4,6,7,3,6,2,4,3,6,5,6,4,0,
    6,6,5,1,4,7,2,1,5,3,4,2,2,7,3,3,6,4,2,4,1,9,7,7,5,2,7,1,7,5,5,1,5,4,1,3,3,4,0,5,1,6,3,5,3,2,3,3,7,2,5,1,1,0,4,1,3,1,0,3,1,6,5,9};

struct kpB_sift { /* SIFT-rank: per interior voxel gradient magnitude + orientation octant, bucketed by octant */
    float mag[NINT + 3];
    unsigned short order[NINT + 3]; /* interior voxels sorted by (octant, raster); packed x | y<<4 | z<<8 */
    unsigned char bin[NINT + 3];
    int start[9];
};
struct kpB_brief { /* BRIEF family: blur output (the middle pass goes back into the patch) */
    float t1[PV + 1];
};
/* LDS of one record: 10.6 KB for both instantiations */
template <bool SIFT>
struct kpB_smem {
    float patch[PV + 1];
    struct {
        typename std::conditional<SIFT, kpB_sift, kpB_brief>::type v;
    } u;
    float sc[16];
    float taps[8];
    float wtab[2][PD + 1];
};

template <bool SIFT>
__global__ __launch_bounds__(DESC_NT) void descriptor_kernel(sift3d_kp_params p, const sift3d_dkp *__restrict__ kps,
                                                        const int *__restrict__ rec_kp, const int *__restrict__ rec_frame,
                                                        long long nrec, sift3d_feature *__restrict__ recs,
                                                        int *__restrict__ rec_group, sift3d_taps taps5)
{
    __shared__ __attribute__((aligned(16))) kpB_smem<SIFT> sm;
    const long long r = xcd_contiguous_item(nrec, p.desc_seg);
    if (r >= nrec) return;
    const int lane = threadIdx.x;
    if (lane < 5) sm.taps[lane] = taps5.f[lane];
    const sift3d_dkp *kp = kps + rec_kp[r];
    const int fr = rec_frame[r]; /* -1: the un-reoriented record */
    float ori[9];
    if (fr < 0) {
        ori[0] = 1; ori[1] = 0; ori[2] = 0; ori[3] = 0; ori[4] = 1; ori[5] = 0; ori[6] = 0; ori[7] = 0; ori[8] = 1;
    } else {
        for (int i = 0; i < 9; i++) ori[i] = kp->frames[fr * 9 + i];
    }
    const sift3d_level lv = p.levels[kp->lvl];
#ifdef SIFT3D_DEV /* timing ablation (tools/desc_ablate.py): every record samples one cache-resident region */
    if (p.debug_stop >= 21 && p.debug_stop <= 26) { /* development aid: every record samples one cache-resident region (22: and runs to the end, 23/24/25: stops where 12/13/14 do, 26: after the bin chains) */
        wave_sample_patch<DESC_NT>(sm.patch, lv.img, lv.X, lv.XP, lv.Y, lv.Z, lv.Zl, lv.z_off, 20.0f, 20.0f, 20.0f, 3.0f, ori);
        if (p.debug_stop == 21) return;
    } else
#endif
    if (fr < 0) {
        /* record 0 is sampled with the identity frame and normalised once inside generateFeature3D
         * (MultiScale.cpp:1742): phase A did exactly that and left the result in patch0 */
        const float *src = p.patch0 + (long long)rec_kp[r] * PV;
        for (int s = lane; s < PV; s += DESC_NT) sm.patch[s] = src[s];
        __syncthreads();
    } else {
        /* The gathers of this phase are paced by the L1's miss handling, and the fewer footprints (a patch's is the size of
         * the whole L1) share a CU's L1 at a time the better it hits -- while the chains of the later phases want every
         * resident workgroup they can hide under.  So at most sampler_cap workgroups per CU sample at once: one takes a
         * token of its CU (a counter in memory, indexed by the hardware's XCC / SE / SH / CU numbers) before the phase and
         * gives it back after it; the others wait asleep and cost no issue slots.  Which CU a counter really belongs to
         * only affects speed. */
        int *tok = nullptr;
        if (p.sampler_cap > 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            tok = p.sampler_tokens + ((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u));
            if (lane == 0) {
                /* (bounded: a counter that was left high -- it cannot be, every taker gives back -- must not hold the kernel
                 * for ever; after ~0.1 s of waiting the workgroup samples without a token of its own) */
                for (int tries = 0;; tries++) {
                    if (atomicAdd(tok, 1) < p.sampler_cap || tries > 50000) break;
                    atomicSub(tok, 1);
                    __builtin_amdgcn_s_sleep(64);
                }
            }
            __syncthreads();
        }
        wave_sample_patch<DESC_NT>(sm.patch, lv.img, lv.X, lv.XP, lv.Y, lv.Z, lv.Zl, lv.z_off, kp->x, kp->y, kp->z, kp->scale, ori);
        if (tok && lane == 0) atomicSub(tok, 1);
    }
    /* ... and every record is normalised once more in main (featExtract.cpp:480) */
#ifdef SIFT3D_DEV
    if (p.debug_stop == 11) return;
#endif
    wave_normalize_patch<DESC_NT>(sm.patch, sm.sc);
#ifdef SIFT3D_DEV
    if (p.debug_stop == 12 || p.debug_stop == 23) return;
#endif

    const bool w0 = lane < 64; /* wavefront 0: lane = descriptor bin for everything that follows the parallel pre-pass */
    float myval = 0.0f;
    if constexpr (SIFT) {
        /* msResampleFeaturesGradientOrientationHistogram, MultiScale.cpp:583-710.  Border voxels
         * have zero gradient (FeatureIO.cpp:2307-2312) and are skipped there, so only the 9^3
         * interior takes part. */
        for (int q = lane; q < NINT; q += DESC_NT) {
            const int x = q % 9 + 1, y = (q / 9) % 9 + 1, z = q / 81 + 1;
            const int s = (z * PD + y) * PD + x;
            float e[3] = {sm.patch[s + 1] - sm.patch[s - 1], sm.patch[s + PD] - sm.patch[s - PD],
                          sm.patch[s + PD * PD] - sm.patch[s - PD * PD]};
            float mg = v3_mag(e);
            int best = 8; /* 8 = no contribution */
            if (mg > 0) {
                v3_norm(e);
                const float oa[8][3] = {{1, 1, 1},  {1, 1, -1},  {1, -1, 1},  {1, -1, -1},
                                        {-1, 1, 1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};
                best = 0;
                float bd = v3_dot(oa[0], e);
#pragma unroll
                for (int t = 1; t < 8; t++) {
                    float d = v3_dot(oa[t], e);
                    if (d > bd) {
                        bd = d;
                        best = t;
                    }
                }
            }
            sm.u.v.mag[q] = mg;
            sm.u.v.bin[q] = (unsigned char)best;
        }
        if (lane < PD) {
            /* spatial coordinate of patch index c in the 2-bin grid (MultiScale.cpp:641-671), then the
             * trilinear weight of bin 0 (wtab[0]) and bin 1 (wtab[1]) along that axis */
            const float binsz = PD / (float)2;
            const int c = lane;
            float v = (int)(c / binsz) + 0.5f;
            if ((int)((c + 0) / binsz) != (int)((c + 1) / binsz)) {
                float p0 = ((c + 0) / binsz);
                float p1 = ((c + 1) / binsz);
                v = (p0 + p1) / 2.0f;
            }
            float w;
            int i0;
            interp_coord(v, 0, 2.0f, i0, w);
            sm.wtab[0][c] = w;
            sm.wtab[1][c] = 1.0f - w;
        }
        __syncthreads();
#ifdef SIFT3D_DEV
        if (p.debug_stop == 13 || p.debug_stop == 24) return;
#endif
        /* bucket the interior voxels by octant, keeping raster order inside a bucket (wavefront 0, ballots) */
        if (w0) {
            int cnt[8];
#pragma unroll
            for (int o = 0; o < 8; o++) cnt[o] = 0;
            for (int base = 0; base < NINT; base += 64) {
                const int q = base + lane;
                const int b = q < NINT ? sm.u.v.bin[q] : 8;
#pragma unroll
                for (int o = 0; o < 8; o++) cnt[o] += __popcll(__ballot(b == o));
            }
            int run[8];
            int acc0 = 0;
#pragma unroll
            for (int o = 0; o < 8; o++) {
                run[o] = acc0;
                if (lane == 0) sm.u.v.start[o] = acc0;
                acc0 += cnt[o];
            }
            if (lane == 0) sm.u.v.start[8] = acc0;
            for (int base = 0; base < NINT; base += 64) {
                const int q = base + lane;
                const int b = q < NINT ? sm.u.v.bin[q] : 8;
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    const unsigned long long m = __ballot(b == o);
                    if (b == o) {
                        const int x = q % 9 + 1, y = (q / 9) % 9 + 1, z = q / 81 + 1;
                        sm.u.v.order[run[o] + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(x | (y << 4) | (z << 8));
                    }
                    run[o] += __popcll(m);
                }
            }
        }
        __syncthreads();
#ifdef SIFT3D_DEV
        if (p.debug_stop == 14 || p.debug_stop == 25) return;
#endif
        if (w0) {
            /* lane = ((z*2+y)*2+x)*8 + orientation: one sequential chain per bin over its octant's voxels */
            const int o = lane & 7, bx = (lane >> 3) & 1, by = (lane >> 4) & 1, bz = (lane >> 5) & 1;
            const float *wxs = sm.wtab[bx], *wys = sm.wtab[by], *wzs = sm.wtab[bz];
            const int k0 = sm.u.v.start[o], k1 = sm.u.v.start[o + 1];
            float acc = 0;
            for (int kk = k0; kk < k1; kk++) {
                const unsigned v = sm.u.v.order[kk];
                const int x = v & 15, y = (v >> 4) & 15, z = v >> 8;
                const float mg = sm.u.v.mag[((z - 1) * 9 + (y - 1)) * 9 + (x - 1)];
                acc += mg * wxs[x] * wys[y] * wzs[z];
            }
#ifdef SIFT3D_DEV
            if (p.debug_stop == 26) return;
#endif
            /* msNormalizeDataPositive, MultiScale.cpp:1580-1611 (lane i's value through v_readlane_b32: a scalar
             * broadcast instead of an LDS permute per term) */
            float mn = 100000;
#pragma unroll
            for (int i = 0; i < 64; i++) {
                float vi = lane_value(acc, i);
                if (vi < mn) mn = vi;
            }
            float v = acc - mn;
            float ss = 0;
#pragma unroll
            for (int i = 0; i < 64; i++) {
                float vi = lane_value(v, i);
                ss += vi * vi;
            }
            float div = 1.0f / sqrtf(ss);
            myval = v * div;
        }
    } else {
        /* msResampleFeaturesBRIEF, MultiScale.cpp:989-1049 */
        wave_blur_patch<DESC_NT>(sm.patch, sm.u.v.t1, sm.patch, sm.taps, 5);
        if (w0) {
            const float *bl = sm.u.v.t1;
            const int x1 = c_brief_x[3 * lane], y1 = c_brief_x[3 * lane + 1], z1 = c_brief_x[3 * lane + 2];
            const int x2 = c_brief_y[3 * lane], y2 = c_brief_y[3 * lane + 1], z2 = c_brief_y[3 * lane + 2];
            float d = bl[x1 + y1 * PD + z1 * PD * PD] - bl[x2 + y2 * PD + z2 * PD * PD];
            if (p.desc_mode == SIFT3D_DESC_BRIEF) {
                myval = (d < 0) ? 1.0f : 0.0f;
            } else if (p.desc_mode == SIFT3D_DESC_RRIEF) {
                myval = d;
            } else {
                float fdx = x1 - x2, fdy = y1 - y2, fdz = z1 - z2;
                int dist = (int)sqrtf(fdx * fdx + fdy * fdy + fdz * fdz);
                myval = d / dist;
            }
        }
    }
    if (!w0) return;
    /* NormalizeDataRankedPCs, MultiScale.cpp:207-233, order of :3148-3176 */
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 64; j++) {
        float vj = lane_value(myval, j);
        rank += (vj < myval || (vj == myval && j < lane)) ? 1 : 0;
    }
    /* where the record goes: slot r of this launch's records -- or, when several contexts write ONE merged list (the slab
     * driver), shifted by a per-group offset: a rank's records are sorted by group already, so the merged position of its
     * record r is r + rec_shift[group] (group = level id * 2 + is_max) */
    const int grp = kp->lvl * 2 + ((kp->info & SIFT3D_INFO_MIN0MAX1) ? 1 : 0);
    sift3d_feature *out = recs + r + (p.rec_shift ? (long long)p.rec_shift[grp] : 0ll);
    out->desc[lane] = (float)rank;
    /* The seventeen words in front of the descriptor, one lane each: one 68-byte store instead of seventeen 4-byte ones by
     * lane 0 (the records may lie in host memory, where every store is a transaction on the bus). */
    static_assert(offsetof(sift3d_feature, desc) == 17 * 4 && offsetof(sift3d_feature, ori) == 16 && offsetof(sift3d_feature, eigs) == 52 &&
                      offsetof(sift3d_feature, info) == 64, "record layout: x y z scale ori[9] eigs[3] info desc[64]");
    if (lane < 17) {
        /* octave -> image space (MultiScale.cpp:531-543), then fSizeFactor (featExtract.cpp:502-505) */
        const float fac = lv.octave_factor, add = 0;
        unsigned word;
        if (lane < 4) {
            float v = lane == 0 ? kp->x : (lane == 1 ? kp->y : (lane == 2 ? kp->z : kp->scale));
            if (lane == 3) v *= fac;
            else v = v * fac + add;
            v *= p.size_factor;
            word = __builtin_bit_cast(unsigned, v);
        } else if (lane < 13) {
            word = __builtin_bit_cast(unsigned, fr < 0 ? kp->ori0[lane - 4] : kp->frames[fr * 9 + lane - 4]);
        } else if (lane < 16) {
            word = __builtin_bit_cast(unsigned, kp->eigs[lane - 13]);
        } else {
            word = kp->info | (fr < 0 ? 0u : SIFT3D_INFO_REORIENT);
        }
        reinterpret_cast<unsigned *>(out)[lane] = word;
        if (lane == 0) rec_group[r] = grp;
    }
}

/* Records per group (level id * 2 + is_max; SIFT3D_GROUPS - 1 takes anything beyond) of a sorted candidate list: what a driver
 * with several contexts needs to place every context's records in one merged list before the descriptor launches run. */
__global__ void group_count_kernel(const unsigned long long *__restrict__ keys, const int *__restrict__ nrec, long long ncand, int *__restrict__ counts)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncand) return;
    const int n = nrec[k];
    if (n <= 0) return;
    int g = (int)(keys[k] >> SIFT3D_KEY_LVL_SHIFT) * 2 + (int)((keys[k] >> SIFT3D_KEY_MAX_SHIFT) & 1ull);
    if (g < 0 || g >= SIFT3D_GROUPS - 1) g = SIFT3D_GROUPS - 1;
    atomicAdd(counts + g, n);
}

hipError_t sift3d_launch_group_counts(hipStream_t s, const unsigned long long *keys, const int *nrec, int64_t ncand, int *counts)
{
    if (ncand <= 0) return hipSuccess;
    hipLaunchKernelGGL(group_count_kernel, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, s, keys, nrec, (long long)ncand, counts);
    return hipGetLastError();
}

/* Record r of candidate k: rec_kp = k (counted over the whole sorted list), rec_frame = -1 (un-reoriented) or the frame
 * index.  The per-keypoint stage runs over the list in chunks: nrec / offs belong to one chunk (offs = exclusive prefix of
 * nrec inside the chunk), its first candidate is cand_off, its first record rec_base[0]; the kernel leaves the first record
 * of the next chunk in rec_base[1]. */
__global__ void recmap_kernel(const int *__restrict__ nrec, const int *__restrict__ offs, long long ncand, int cand_off,
                              int *rec_base, int *__restrict__ rec_kp, int *__restrict__ rec_frame, unsigned long long *kp_count)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncand) return;
    const int n = nrec[k], o = rec_base[0] + offs[k];
    if (k == ncand - 1) rec_base[1] = o + n;
    if (n > 0) atomicAdd(kp_count, 1ull); /* keypoints that survived the bounds and eigenvalue tests (statistics) */
    for (int f = 0; f < n; f++) {
        rec_kp[o + f] = cand_off + (int)k;
        rec_frame[o + f] = f - 1;
    }
}

/* ---------------------------------------------------------------------- */
/* launchers                                                               */
/* ---------------------------------------------------------------------- */
hipError_t sift3d_launch_keypointsA(hipStream_t s, const sift3d_kp_params &p, const unsigned long long *keys,
                                    const sift3d_cval *vals, int64_t ncand, sift3d_dkp *kps, int *nrec, const float *taps3)
{
    if (ncand <= 0) return hipSuccess;
    sift3d_taps t;
    for (int i = 0; i < 17; i++) t.f[i] = i < 3 ? taps3[i] : 0.0f;
    hipLaunchKernelGGL(keypoint_kernel, dim3((unsigned)ncand), dim3(KP_NT), 0, s, p, keys, vals, (long long)ncand, kps, nrec, t);
    return hipGetLastError();
}

hipError_t sift3d_launch_recmap(hipStream_t s, const int *nrec, const int *offs, int64_t ncand, int cand_off, int *rec_base,
                                int *rec_kp, int *rec_frame, unsigned long long *kp_count)
{
    if (ncand <= 0) return hipSuccess;
    hipLaunchKernelGGL(recmap_kernel, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, s, nrec, offs, (long long)ncand,
                       cand_off, rec_base, rec_kp, rec_frame, kp_count);
    return hipGetLastError();
}

hipError_t sift3d_launch_descriptors(hipStream_t s, const sift3d_kp_params &p, const sift3d_dkp *kps, const int *rec_kp,
                                     const int *rec_frame, int64_t nrec, sift3d_feature *recs, int *rec_group,
                                     const float *taps5)
{
    if (nrec <= 0) return hipSuccess;
    sift3d_taps t;
    for (int i = 0; i < 17; i++) t.f[i] = i < 5 ? taps5[i] : 0.0f;
    const int seg = p.desc_seg;
    const dim3 grid((unsigned)(seg > 0 ? (nrec + seg - 1) / seg * seg : ((nrec + 7) / 8) * 8));
    if (p.desc_mode == SIFT3D_DESC_SIFT)
        hipLaunchKernelGGL(descriptor_kernel<true>, grid, dim3(DESC_NT), 0, s, p, kps, rec_kp, rec_frame, (long long)nrec, recs,
                           rec_group, t);
    else
        hipLaunchKernelGGL(descriptor_kernel<false>, grid, dim3(DESC_NT), 0, s, p, kps, rec_kp, rec_frame, (long long)nrec, recs,
                           rec_group, t);
    return hipGetLastError();
}
