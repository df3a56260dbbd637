/*
 * sift3d_oracle.c -- CPU restatement of the reference featExtract path.
 *
 * TEST INFRASTRUCTURE ONLY (see sift3d_oracle.h).  Plain C, single thread,
 * 64-bit indexing.  Written from the behaviour of the reference, not from its
 * text: loops are organised for the cache, but every floating-point operation
 * happens in the same type and the same order as in the cited reference lines
 * (float vs double, separate multiply and add, ascending taps, raster order).
 *
 * The reference is C++: <math.h> there resolves sqrt/exp/fabs/floor on a
 * float argument to the float overloads (nm of its GaussianMask.o shows expf).
 * Each libm call below is therefore spelled with the width the C++ overload
 * resolution picks.
 *
 * R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/
 */
#include "sift3d_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PD O3_PATCH_DIM
#define PV O3_PATCH_VOX

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void o3_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* Gaussian taps                                                            */
/* ------------------------------------------------------------------------ */

/* R/src_common/GaussianMask.cpp:12-57 calculate_gaussian_filter_size */
/* O3_REFBIN_VARIANT (test-only second build, oracle/_build/libsift3d_oracle_refbin.so): the arithmetic of the CPU binary
 * the reference repository ships (R/bin/Linux/featExtract, GCC 5.4, not stripped, no FMA), read from its disassembly
 * (objdump -d, never executed) wherever it differs from what a current g++ makes of the same source lines.  Every such
 * place is one #ifdef below with the address of the instructions that show it.  tests/test_oracle_pins.py holds this
 * build to the binary's own .key files (tests/golden/refbin_*.key). */
#ifdef O3_REFBIN_VARIANT
/* GaussianMask.cpp includes <math.h> without <cmath>'s overloads in that toolchain: exp(float) is the C function
 * exp(double).  0x451c87-0x451c90 and 0x451d0e-0x451d17: cvtss2sd, call exp@plt, cvtsd2ss. */
#define O3_EXPF(x) ((float)exp((double)(x)))
#else
#define O3_EXPF(x) expf(x)
#endif
int o3_gauss_filter_size(float sigma, float min_value)
{
    float power = 0.0f;
    float value = O3_EXPF(power);
    int i;
    if (sigma == 0) return 1;
    float cur = 1, nxt = 1;
    i = 0;
    do {
        i++;
        cur = nxt;
        power = ((float)(i * i)) / ((float)-2.0 * sigma * sigma);
        nxt = cur + 2 * O3_EXPF(power);
    } while (nxt - cur > 0.00001f);
    for (i = 1; value <= cur * (1.0f - min_value); i++) {
        power = ((float)(i * i)) / ((float)-2.0 * sigma * sigma);
        value += 2 * O3_EXPF(power);
    }
    i--;
    return 2 * i + 1;
}

/* R/src_common/GaussianMask.cpp:241-265 generate_gaussian_filter1d, then the
 * normalisation of R/src_common/GaussBlur3D.cpp:1190-1201 */
int o3_gauss_taps_raw(float sigma, float min_value, float *taps, int normalise)
{
    int n = o3_gauss_filter_size(sigma, min_value);
    if (n > O3_MAX_TAPS) return -1;
    if (sigma > 0.0f) {
        const double PI_ = 3.1415926535897932384626433832795;
        float mean = (float)(n / 2);
        float sig2 = sigma * sigma;
        float scale = (float)(1.0 / (sigma * sqrt(2.0 * PI_)));
        for (int j = 0; j < n; j++) {
            float pos = ((float)j - mean);
            float power = ((pos * pos) / sig2) / (float)(-2.0f);
#ifdef O3_REFBIN_VARIANT
            /* 0x4523b7-0x4523ca (generate_gaussian_filter1d(PpImage&, ...)): cvtss2sd power; call exp@plt; mulsd by the
             * scale (rounded to float at 0x45237d and widened again at 0x452381); cvtsd2ss -- the product is formed in
             * double and rounded once */
            taps[j] = (float)((double)scale * exp((double)power));
#else
            taps[j] = (float)(scale * expf(power));
#endif
        }
    } else {
        taps[0] = 1;
    }
    if (!normalise) return n;
    float sum = 0;
    for (int c = 0; c < n; c++) sum += taps[c];
    for (int c = 0; c < n; c++) taps[c] /= sum;
    return n;
}

int o3_gauss_taps(float sigma, float min_value, float *taps) { return o3_gauss_taps_raw(sigma, min_value, taps, 1); }

/* ------------------------------------------------------------------------ */
/* Separable blur: R/src_common/GaussBlur3D.cpp:43-61 (filter_1d) and        */
/* :329-479 (blur_3d_simpleborders): passes x, y, z; float intermediates;    */
/* zero outside the volume; out = sum_{j ascending} f[j]*in[c+j-half],       */
/* accumulator starting at 0, multiply and add rounded separately.           */
/* Taps that fall outside the volume contribute f*0 = +0, which leaves a     */
/* float accumulator unchanged, so they are skipped.                         */
/* ------------------------------------------------------------------------ */
void o3_filter3d(const float *in, float *out, int64_t X, int64_t Y, int64_t Z, const float *taps, int n)
{
    const int h = n / 2;
    const int64_t XY = X * Y;
    /* two volume-sized temporaries, kept between calls: re-faulting 2N floats per blur dominated the run time.
     * Thread-local: the OpenMP build (the multi-core CPU baseline, see the Makefile) also calls this function for the
     * 11^3 orientation histograms from inside its per-keypoint parallel loop. */
    static _Thread_local float *ws = 0;
    static _Thread_local size_t ws_n = 0;
    if ((size_t)(2 * XY * Z) > ws_n) {
        free(ws);
        ws_n = (size_t)(2 * XY * Z);
        ws = (float *)malloc(sizeof(float) * ws_n);
    }
    float *t1 = ws, *t2 = ws + XY * Z;
    /* Every output row is computed by one thread with the serial code's arithmetic, so the OpenMP build is bit-identical
     * to the serial one (tests/test_oracle_pins.py checks it).  Inside a parallel region (patch-sized calls) the inner
     * region runs on the calling thread alone. */
#ifdef _OPENMP
#pragma omp parallel if (XY * Z >= 32768)
#endif
    {
        float *line = (float *)calloc((size_t)(X + n), sizeof(float));
        float *acc = (float *)malloc(sizeof(float) * (size_t)X);

        /* x pass */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t r = 0; r < Y * Z; r++) {
            const float *src = in + r * X;
            float *dst = t1 + r * X;
            memcpy(line + h, src, sizeof(float) * (size_t)X);
            for (int64_t c = 0; c < X; c++) {
                float s = 0;
                for (int j = 0; j < n; j++) s += taps[j] * line[c + j];
                dst[c] = s;
            }
        }
        /* y pass */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t z = 0; z < Z; z++) {
            for (int64_t y = 0; y < Y; y++) {
                for (int64_t x = 0; x < X; x++) acc[x] = 0;
                for (int j = 0; j < n; j++) {
                    int64_t yy = y + j - h;
                    if (yy < 0 || yy >= Y) continue;
                    const float *src = t1 + z * XY + yy * X;
                    const float f = taps[j];
                    for (int64_t x = 0; x < X; x++) acc[x] += f * src[x];
                }
                memcpy(t2 + z * XY + y * X, acc, sizeof(float) * (size_t)X);
            }
        }
        /* z pass */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t z = 0; z < Z; z++) {
            for (int64_t y = 0; y < Y; y++) {
                for (int64_t x = 0; x < X; x++) acc[x] = 0;
                for (int j = 0; j < n; j++) {
                    int64_t zz = z + j - h;
                    if (zz < 0 || zz >= Z) continue;
                    const float *src = t2 + zz * XY + y * X;
                    const float f = taps[j];
                    for (int64_t x = 0; x < X; x++) acc[x] += f * src[x];
                }
                memcpy(out + z * XY + y * X, acc, sizeof(float) * (size_t)X);
            }
        }
        free(acc);
        free(line);
    }
}

/* R/src_common/GaussBlur3D.cpp:1159-1258 gb3d_blur3d_interleave (CPU branch, z > 1) */
int o3_blur(const float *in, float *out, int64_t X, int64_t Y, int64_t Z, float sigma, float min_value)
{
    float taps[O3_MAX_TAPS];
    int n = o3_gauss_taps(sigma, min_value, taps);
    if (n < 0) return 0;
    o3_filter3d(in, out, X, Y, Z, taps, n);
    return 1;
}

/* R/src_common/FeatureIO.cpp:1950-1987 fioMultSum with fMultIn2 = -1.0f */
void o3_dog(const float *a, const float *b, float *out, int64_t n)
{
    const float m = -1.0f;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (n >= 32768)
#endif
    for (int64_t i = 0; i < n; i++) out[i] = a[i] + m * b[i];
}

/* R/src_common/FeatureIO.cpp:1474-1554 fioSubSampleInterpolate */
void o3_subsample(const float *in, int64_t X, int64_t Y, int64_t Z, float *out)
{
    const int64_t ox = X / 2, oy = Y / 2, oz = Z / 2;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (ox * oy * oz >= 32768)
#endif
    for (int64_t z = 0; z < oz; z++)
        for (int64_t y = 0; y < oy; y++)
            for (int64_t x = 0; x < ox; x++) {
                const float *p0 = in + ((2 * z) * Y + 2 * y) * X + 2 * x;
                const float *p1 = p0 + X * Y;
                float s = 0;
                s += p0[0] + p0[X] + p0[1] + p0[X + 1];
                if (2 * z + 1 < Z) {
                    s += p1[0] + p1[X] + p1[1] + p1[X + 1];
                    s *= 0.125;
                } else {
                    s *= 0.25;
                }
                out[(z * oy + y) * ox + x] = s;
            }
}

/* R/src_common/FeatureIO.cpp:2452-2548 fioDoubleSize (out is 2X x 2Y x 2Z) */
void o3_double_size(const float *in, int64_t X, int64_t Y, int64_t Z, float *out)
{
    const int64_t DX = X > 1 ? 2 * X : X, DY = Y > 1 ? 2 * Y : Y, DZ = Z > 1 ? 2 * Z : Z;
    for (int64_t z = 0; z < Z; z++)
        for (int64_t y = 0; y < Y; y++)
            for (int64_t x = 0; x < X; x++) {
                float lo[2][2][2];
                for (int zz = 0; zz <= 1; zz++) {
                    int dz = zz;
                    if (z + zz >= Z) dz = 0;
                    for (int yy = 0; yy <= 1; yy++) {
                        int dy = yy;
                        if (y + yy >= Y) dy = 0;
                        for (int xx = 0; xx <= 1; xx++) {
                            int dx = xx;
                            if (x + xx >= X) dx = 0;
                            lo[zz][yy][xx] = in[((z + dz) * Y + (y + dy)) * X + (x + dx)];
                        }
                    }
                }
                float hi[2][2][2];
                hi[0][0][0] = lo[0][0][0];
                hi[1][0][0] = 0.5f * (lo[0][0][0] + lo[1][0][0]);
                hi[0][1][0] = 0.5f * (lo[0][0][0] + lo[0][1][0]);
                hi[0][0][1] = 0.5f * (lo[0][0][0] + lo[0][0][1]);
                hi[1][1][0] = 0.25f * (lo[0][0][0] + lo[1][0][0] + lo[0][1][0] + lo[1][1][0]);
                hi[0][1][1] = 0.25f * (lo[0][0][0] + lo[0][1][0] + lo[0][0][1] + lo[0][1][1]);
                hi[1][0][1] = 0.25f * (lo[0][0][0] + lo[1][0][0] + lo[0][0][1] + lo[1][0][1]);
                hi[1][1][1] = 0.125f * (lo[0][0][0] + lo[0][0][1] + lo[0][1][0] + lo[0][1][1] + lo[1][0][0] +
                                        lo[1][0][1] + lo[1][1][0] + lo[1][1][1]);
                for (int zz = 0; zz <= 1; zz++) {
                    int dz = zz;
                    if (2 * z + zz >= DZ) dz = 0;
                    for (int yy = 0; yy <= 1; yy++) {
                        int dy = yy;
                        if (2 * y + yy >= DY) dy = 0;
                        for (int xx = 0; xx <= 1; xx++) {
                            int dx = xx;
                            if (2 * x + xx >= DX) dx = 0;
                            out[((2 * z + dz) * DY + (2 * y + dy)) * DX + (2 * x + dx)] = hi[dz][dy][dx];
                        }
                    }
                }
            }
}

/* R/src_common/FeatureIO.cpp:1670-1714 fioSubSample2DCenterPixel (out is X/2 x Y/2 x Z/2) */
void o3_halve_center(const float *in, int64_t X, int64_t Y, int64_t Z, float *out)
{
    const int64_t ox = X / 2, oy = Y / 2, oz = Z / 2;
    for (int64_t z = 0; z < oz; z++)
        for (int64_t y = 0; y < oy; y++)
            for (int64_t x = 0; x < ox; x++) {
                float v = 0;
                v += in[((2 * z + 0) * Y + 2 * y + 0) * X + 2 * x + 0];
                v += in[((2 * z + 1) * Y + 2 * y + 0) * X + 2 * x + 0];
                v += in[((2 * z + 0) * Y + 2 * y + 1) * X + 2 * x + 0];
                v += in[((2 * z + 1) * Y + 2 * y + 1) * X + 2 * x + 0];
                v += in[((2 * z + 0) * Y + 2 * y + 0) * X + 2 * x + 1];
                v += in[((2 * z + 1) * Y + 2 * y + 0) * X + 2 * x + 1];
                v += in[((2 * z + 0) * Y + 2 * y + 1) * X + 2 * x + 1];
                v += in[((2 * z + 1) * Y + 2 * y + 1) * X + 2 * x + 1];
                out[(z * oy + y) * ox + x] = v / 8.0f;
            }
}

/* ------------------------------------------------------------------------ */
/* Extrema                                                                  */
/* ------------------------------------------------------------------------ */

/* neighbour offsets in the order of R/src_common/MultiScale.cpp:2282-2309 */
static void nbr_offsets(int64_t X, int64_t Y, int64_t off[26])
{
    int k = 0;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (dz == 0 && dy == 0 && dx == 0) continue;
                off[k++] = dz * X * Y + dy * X + dx;
            }
}

/* R/src_common/MultiScale.cpp:2260-2400 regFindFEATUREIO + :2408-2524
 * peakFunction4D / valleyFunction4D with pfioL == NULL: strict extremum over
 * the 26 neighbours in C, then centre + 26 in H.  Raster z,y,x order. */
/* The per-voxel test of the scan: +1 = maximum, -1 = minimum, 0 = neither (strict compares, as there). */
static inline int detect_voxel(const float *H, const float *C, int64_t idx, const int64_t off[26])
{
    const float c = C[idx];
    /* sign of (c - first neighbour) decides which test can still pass */
    const float d0 = c - C[idx + off[0]];
    int sg = (0.0f < d0) - (d0 < 0.0f);
    if (sg == 0) return 0;
    int ok = 1;
    for (int n = 1; n < 26 && ok; n++) {
        const float d = c - C[idx + off[n]];
        int s = (0.0f < d) - (d < 0.0f);
        ok &= (s == sg);
    }
    if (!ok) return 0;
    if (sg > 0) { /* peak: all of H (centre + 26) strictly lower */
        int p = (H[idx] < c);
        for (int n = 0; n < 26 && p; n++) p &= (H[idx + off[n]] < c);
        return p ? 1 : 0;
    }
    int p = (H[idx] > c);
    for (int n = 0; n < 26 && p; n++) p &= (H[idx + off[n]] > c);
    return p ? -1 : 0;
}

#ifdef _OPENMP
typedef struct {
    o3_extremum *v;
    int64_t n, cap;
} extvec;
static void ev_push(extvec *e, int x, int y, int z, float c)
{
    if (e->n == e->cap) {
        e->cap = e->cap ? 2 * e->cap : 1024;
        e->v = (o3_extremum *)realloc(e->v, sizeof(o3_extremum) * (size_t)e->cap);
    }
    e->v[e->n].x = x; e->v[e->n].y = y; e->v[e->n].z = z; e->v[e->n].value = c;
    e->n++;
}
#endif

int o3_detect(const float *H, const float *C, int64_t X, int64_t Y, int64_t Z,
              o3_extremum *minima, int64_t cap_min, int64_t *n_min,
              o3_extremum *maxima, int64_t cap_max, int64_t *n_max)
{
    int64_t off[26];
    nbr_offsets(X, Y, off);
    int64_t nmin = 0, nmax = 0;
    int overflow = 0;
#ifdef _OPENMP
    if (X * Y * Z >= 32768) {
        /* every thread scans a contiguous range of planes into lists of its own; ranges ascend with the thread number,
         * so concatenating the lists in thread order is the serial raster order */
        const int nt = omp_get_max_threads();
        extvec *lmin = (extvec *)calloc((size_t)nt, sizeof(extvec)), *lmax = (extvec *)calloc((size_t)nt, sizeof(extvec));
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num(), n = omp_get_num_threads();
            const int64_t planes = Z - 2 > 0 ? Z - 2 : 0;
            const int64_t z0 = 1 + planes * t / n, z1 = 1 + planes * (t + 1) / n;
            for (int64_t z = z0; z < z1; z++)
                for (int64_t y = 1; y < Y - 1; y++)
                    for (int64_t x = 1; x < X - 1; x++) {
                        const int64_t idx = (z * Y + y) * X + x;
                        const int k = detect_voxel(H, C, idx, off);
                        if (k > 0) ev_push(&lmax[t], (int)x, (int)y, (int)z, C[idx]);
                        else if (k < 0) ev_push(&lmin[t], (int)x, (int)y, (int)z, C[idx]);
                    }
        }
        for (int t = 0; t < nt; t++) {
            for (int64_t i = 0; i < lmin[t].n; i++, nmin++)
                if (nmin < cap_min) minima[nmin] = lmin[t].v[i]; else overflow = 1;
            for (int64_t i = 0; i < lmax[t].n; i++, nmax++)
                if (nmax < cap_max) maxima[nmax] = lmax[t].v[i]; else overflow = 1;
            free(lmin[t].v); free(lmax[t].v);
        }
        free(lmin); free(lmax);
        *n_min = nmin;
        *n_max = nmax;
        return overflow ? -1 : 0;
    }
#endif
    for (int64_t z = 1; z < Z - 1; z++)
        for (int64_t y = 1; y < Y - 1; y++)
            for (int64_t x = 1; x < X - 1; x++) {
                const int64_t idx = (z * Y + y) * X + x;
                const int k = detect_voxel(H, C, idx, off);
                if (k > 0) {
                    if (nmax < cap_max) {
                        maxima[nmax].x = (int)x; maxima[nmax].y = (int)y; maxima[nmax].z = (int)z;
                        maxima[nmax].value = C[idx];
                    } else overflow = 1;
                    nmax++;
                } else if (k < 0) {
                    if (nmin < cap_min) {
                        minima[nmin].x = (int)x; minima[nmin].y = (int)y; minima[nmin].z = (int)z;
                        minima[nmin].value = C[idx];
                    } else overflow = 1;
                    nmin++;
                }
            }
    *n_min = nmin;
    *n_max = nmax;
    return overflow ? -1 : 0;
}

/* R/src_common/MultiScale.cpp:1135-1223 validateDifferencePeak3D */
int o3_validate_peak(const o3_extremum *e, const float *G1, const float *G2, int64_t X, int64_t Y, int64_t Z)
{
    (void)Z;
    int64_t off[26];
    nbr_offsets(X, Y, off);
    const int64_t idx = ((int64_t)e->z * Y + e->y) * X + e->x;
    const float c = e->value;
    float v = G1[idx] - G2[idx];
    int p = (v < c);
    for (int n = 0; n < 26 && p; n++) {
        v = G1[idx + off[n]] - G2[idx + off[n]];
        p &= (v < c);
    }
    return p;
}

/* R/src_common/MultiScale.cpp:1230-1318 validateDifferenceValley3D */
int o3_validate_valley(const o3_extremum *e, const float *G1, const float *G2, int64_t X, int64_t Y, int64_t Z)
{
    (void)Z;
    int64_t off[26];
    nbr_offsets(X, Y, off);
    const int64_t idx = ((int64_t)e->z * Y + e->y) * X + e->x;
    const float c = e->value;
    float v = G1[idx] - G2[idx];
    int p = (v > c);
    for (int n = 0; n < 26 && p; n++) {
        v = G1[idx + off[n]] - G2[idx + off[n]];
        p &= (v > c);
    }
    return p;
}

/* The same decision as o3_detect followed by o3_validate_* when the next DoG
 * is stored (Dnext == G1 - G2 exactly): 26 + 27 + 27 strict comparisons. */
int o3_detect3(const float *Dp, const float *Dc, const float *Dn, int64_t X, int64_t Y, int64_t Z,
               o3_extremum *minima, int64_t cap_min, int64_t *n_min,
               o3_extremum *maxima, int64_t cap_max, int64_t *n_max)
{
    int64_t off[26];
    nbr_offsets(X, Y, off);
    int64_t nmin = 0, nmax = 0;
    int overflow = 0;
    for (int64_t z = 1; z < Z - 1; z++)
        for (int64_t y = 1; y < Y - 1; y++)
            for (int64_t x = 1; x < X - 1; x++) {
                const int64_t idx = (z * Y + y) * X + x;
                const float c = Dc[idx];
                int mx = 1, mn = 1;
                for (int n = 0; n < 26 && (mx | mn); n++) {
                    const float v = Dc[idx + off[n]];
                    mx &= (v < c);
                    mn &= (v > c);
                }
                if (!(mx | mn)) continue;
                const float *lv[2] = {Dp, Dn};
                for (int l = 0; l < 2 && (mx | mn); l++) {
                    float v = lv[l][idx];
                    mx &= (v < c);
                    mn &= (v > c);
                    for (int n = 0; n < 26 && (mx | mn); n++) {
                        v = lv[l][idx + off[n]];
                        mx &= (v < c);
                        mn &= (v > c);
                    }
                }
                if (mx) {
                    if (nmax < cap_max) {
                        maxima[nmax].x = (int)x; maxima[nmax].y = (int)y; maxima[nmax].z = (int)z;
                        maxima[nmax].value = c;
                    } else overflow = 1;
                    nmax++;
                } else if (mn) {
                    if (nmin < cap_min) {
                        minima[nmin].x = (int)x; minima[nmin].y = (int)y; minima[nmin].z = (int)z;
                        minima[nmin].value = c;
                    } else overflow = 1;
                    nmin++;
                }
            }
    *n_min = nmin;
    *n_max = nmax;
    return overflow ? -1 : 0;
}

/* ------------------------------------------------------------------------ */
/* Sub-voxel refinement                                                     */
/* ------------------------------------------------------------------------ */

/* R/src_common/MultiScale.cpp:2531-2534 finddet */
static double finddet(double a1, double a2, double a3, double b1, double b2, double b3, double c1, double c2, double c3)
{
    return ((a1 * b2 * c3) - (a1 * b3 * c2) - (a2 * b1 * c3) + (a3 * b1 * c2) + (a2 * b3 * c1) - (a3 * b2 * c1));
}

/* R/src_common/MultiScale.cpp:1641-1697 interpolate_extremum_quadratic
 * (diagnostic printf calls dropped; every non-returning branch falls through
 * to "return x1" exactly as there) */
double o3_interp_quadratic(double x0, double x1, double x2, double fx0, double fx1, double fx2)
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

/* R/src_common/MultiScale.cpp:1614-1639 interpolate_discrete_3D_point */
void o3_interp_point(const float *C, int64_t X, int64_t Y, int64_t Z, int ix, int iy, int iz, float *fx, float *fy, float *fz)
{
    (void)Z;
    const int64_t idx = ((int64_t)iz * Y + iy) * X + ix;
    const float c = C[idx];
    *fx = (float)o3_interp_quadratic(ix - 1, ix, ix + 1, C[idx - 1], c, C[idx + 1]);
    *fy = (float)o3_interp_quadratic(iy - 1, iy, iy + 1, C[idx - X], c, C[idx + X]);
    *fz = (float)o3_interp_quadratic(iz - 1, iz, iz + 1, C[idx - X * Y], c, C[idx + X * Y]);
}

/* ------------------------------------------------------------------------ */
/* Patch sampling                                                           */
/* ------------------------------------------------------------------------ */

/* R/src_common/FeatureIO.cpp:757-782 _fioDetermineInterpCoord */
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

/* R/src_common/FeatureIO.cpp:812-850 fioGetPixelTrilinearInterp */
float o3_trilinear(const float *img, int64_t X, int64_t Y, int64_t Z, float x, float y, float z)
{
    float wx, wy, wz;
    int ix, iy, iz;
    interp_coord(x, 0, (float)X, &ix, &wx);
    interp_coord(y, 0, (float)Y, &iy, &wy);
    interp_coord(z, 0, (float)Z, &iz, &wz);
    const float *p = img + ((int64_t)iz * Y + iy) * X + ix;
    const int64_t XY = X * Y;
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

/* R/src_common/FeatureIO.cpp:853-889 fioIncPixelTrilinearInterp on a dense
 * grid of dims (gx,gy,gz) with nf interleaved features */
static void splat(float *grid, int gx, int gy, int gz, int nf, float x, float y, float z, int feat, float value)
{
    float wx, wy, wz;
    int ix, iy, iz;
    interp_coord(x, 0, (float)gx, &ix, &wx);
    interp_coord(y, 0, (float)gy, &iy, &wy);
    interp_coord(z, 0, (float)gz, &iz, &wz);
#define G_(a, b, c) grid[((((int64_t)(iz + (c))) * gy + (iy + (b))) * gx + (ix + (a))) * nf + feat]
    G_(0, 0, 0) += value * wx * wy * wz;
    G_(1, 0, 0) += value * (1.0f - wx) * wy * wz;
    G_(0, 1, 0) += value * wx * (1.0f - wy) * wz;
    G_(1, 1, 0) += value * (1.0f - wx) * (1.0f - wy) * wz;
    G_(0, 0, 1) += value * wx * wy * (1.0f - wz);
    G_(1, 0, 1) += value * (1.0f - wx) * wy * (1.0f - wz);
    G_(0, 1, 1) += value * wx * (1.0f - wy) * (1.0f - wz);
    G_(1, 1, 1) += value * (1.0f - wx) * (1.0f - wy) * (1.0f - wz);
#undef G_
}

/* R/src_common/MultiScale.h:192-222 invert_3x3<float,double> */
void o3_invert3(float in[3][3], float out[3][3])
{
    float a11 = in[0][0], a21 = in[1][0], a31 = in[2][0];
    float a12 = in[0][1], a22 = in[1][1], a32 = in[2][1];
    float a13 = in[0][2], a23 = in[1][2], a33 = in[2][2];
    float det = a11 * (a33 * a22 - a32 * a23) - a21 * (a33 * a12 - a32 * a13) + a31 * (a23 * a12 - a22 * a13);
    double div = 1 / (double)det;
    out[0][0] = (float)((a33 * a22 - a32 * a23) * div);
    out[1][0] = (float)(-(a33 * a21 - a31 * a23) * div);
    out[2][0] = (float)((a32 * a21 - a31 * a22) * div);
    out[0][1] = (float)(-(a33 * a12 - a32 * a13) * div);
    out[1][1] = (float)((a33 * a11 - a31 * a13) * div);
    out[2][1] = (float)(-(a32 * a11 - a31 * a12) * div);
    out[0][2] = (float)((a23 * a12 - a22 * a13) * div);
    out[1][2] = (float)(-(a23 * a11 - a21 * a13) * div);
    out[2][2] = (float)((a22 * a11 - a21 * a12) * div);
}

/* R/src_common/MultiScale.cpp:2614-2714 sampleImage3D.  Returns 0, or -1 when
 * the keypoint is too close to the border. */
int o3_sample_patch(const o3_feature *f, const float *img, int64_t X, int64_t Y, int64_t Z, float *patch)
{
    float inv[3][3];
    float rad = 2.0f * f->scale;
    int rmax = (int)(rad + 2);
    if (f->x - rmax < 0 || f->y - rmax < 0 || f->z - rmax < 0 || f->x + rmax >= X || f->y + rmax >= Y ||
        f->z + rmax >= Z)
        return -1;
    float ori[3][3];
    memcpy(ori, f->ori, sizeof(ori));
    o3_invert3(ori, inv);
    const int sr = PD / 2;
    for (int z = -sr; z <= sr; z++)
        for (int y = -sr; y <= sr; y++)
            for (int x = -sr; x <= sr; x++) {
                float in3[3] = {(float)x, (float)y, (float)z};
                float o[3];
                for (int i = 0; i < 3; i++) { /* MultiScale.h:494-510 mult_3x3<float,double> */
                    o[i] = 0;
                    for (int j = 0; j < 3; j++) o[i] += inv[i][j] * in3[j];
                }
                float sc = rad / (float)(sr);
                o[0] *= sc; o[1] *= sc; o[2] *= sc;
                o[0] += f->x; o[1] += f->y; o[2] += f->z;
                float pix;
                if (o[0] < 0 || o[0] >= X) /* the reference tests x three times, :2687-2689 */
                    pix = 0;
                else
                    pix = o3_trilinear(img, X, Y, Z, o[0], o[1], o[2]);
                patch[((z + sr) * PD + (y + sr)) * PD + (x + sr)] = pix;
            }
    return 0;
}

/* R/src_common/MultiScale.cpp:127-205 Feature3D::NormalizeData */
void o3_normalize_patch(float *d)
{
    float sum = 0;
    for (int i = 0; i < PV; i++) sum += d[i];
    float mean = sum / (PD * PD * PD);
    float ss = 0;
    for (int i = 0; i < PV; i++) {
        d[i] -= mean;
        ss += d[i] * d[i];
    }
    float div = 1.0f / sqrtf(ss);
    for (int i = 0; i < PV; i++) d[i] *= div;
}

/* R/src_common/FeatureIO.cpp:2284-2326 fioGenerateEdgeImages3D on the 11^3 patch */
static void patch_edges(const float *d, float *dx, float *dy, float *dz)
{
    memset(dx, 0, sizeof(float) * PV);
    memset(dy, 0, sizeof(float) * PV);
    memset(dz, 0, sizeof(float) * PV);
    for (int z = 1; z < PD - 1; z++)
        for (int y = 1; y < PD - 1; y++)
            for (int x = 1; x < PD - 1; x++) {
                int i = (z * PD + y) * PD + x;
                dx[i] = d[i + 1] - d[i - 1];
                dy[i] = d[i + PD] - d[i - PD];
                dz[i] = d[i + PD * PD] - d[i - PD * PD];
            }
}

/* ------------------------------------------------------------------------ */
/* 3x3 SVD: R/src_common/SVD.h:44-228 SingularValueDecomp<float,3,3> (the     */
/* Numerical Recipes svdcmp with double temporaries and float storage), and  */
/* :15-31 SortEigenDecomp.                                                   */
/* ------------------------------------------------------------------------ */
#define SVD_SIGN(a, b) ((b) >= 0.0 ? fabs(a) : -fabs(a))
#define SVD_PYTHAG(a, b) (sqrt((a) * (a) + (b) * (b)))
void o3_svd3(float mat[3][3], float w[3], float v[3][3])
{
    const int m = 3, n = 3;
    int flag, i, its, j, jj, k, l = 0, nm = 0;
    double anorm, c, f, g, h, s, scale, x, y, z;
    double rv1[3];
    g = scale = anorm = 0.0;
    for (i = 1; i <= n; i++) {
        l = i + 1;
        rv1[i - 1] = scale * g;
        g = s = scale = 0.0;
        if (i <= m) {
            for (k = i; k <= m; k++) scale += fabsf(mat[k - 1][i - 1]);
            if (scale) {
                for (k = i; k <= m; k++) {
                    mat[k - 1][i - 1] = (float)(mat[k - 1][i - 1] / scale);
                    s += mat[k - 1][i - 1] * mat[k - 1][i - 1];
                }
                f = mat[i - 1][i - 1];
                g = -SVD_SIGN(sqrt(s), f);
                h = f * g - s;
                mat[i - 1][i - 1] = (float)(f - g);
                for (j = l; j <= n; j++) {
                    for (s = 0.0, k = i; k <= m; k++) s += mat[k - 1][i - 1] * mat[k - 1][j - 1];
                    f = s / h;
                    for (k = i; k <= m; k++) mat[k - 1][j - 1] = (float)(mat[k - 1][j - 1] + f * mat[k - 1][i - 1]);
                }
                for (k = i; k <= m; k++) mat[k - 1][i - 1] = (float)(mat[k - 1][i - 1] * scale);
            }
        }
        w[i - 1] = (float)(scale * g);
        g = s = scale = 0.0;
        if (i <= m && i != n) {
            for (k = l; k <= n; k++) scale += fabsf(mat[i - 1][k - 1]);
            if (scale) {
                for (k = l; k <= n; k++) {
                    mat[i - 1][k - 1] = (float)(mat[i - 1][k - 1] / scale);
                    s += mat[i - 1][k - 1] * mat[i - 1][k - 1];
                }
                f = mat[i - 1][l - 1];
                g = -SVD_SIGN(sqrt(s), f);
                h = f * g - s;
                mat[i - 1][l - 1] = (float)(f - g);
                for (k = l; k <= n; k++) rv1[k - 1] = mat[i - 1][k - 1] / h;
                for (j = l; j <= m; j++) {
                    for (s = 0.0, k = l; k <= n; k++) s += mat[j - 1][k - 1] * mat[i - 1][k - 1];
                    for (k = l; k <= n; k++) mat[j - 1][k - 1] = (float)(mat[j - 1][k - 1] + s * rv1[k - 1]);
                }
                for (k = l; k <= n; k++) mat[i - 1][k - 1] = (float)(mat[i - 1][k - 1] * scale);
            }
        }
        {
            double t = (fabsf(w[i - 1]) + fabs(rv1[i - 1]));
            anorm = (anorm > t ? anorm : t);
        }
    }
    for (i = n; i >= 1; i--) {
        if (i < n) {
            if (g) {
                for (j = l; j <= n; j++) v[j - 1][i - 1] = (float)((mat[i - 1][j - 1] / mat[i - 1][l - 1]) / g);
                for (j = l; j <= n; j++) {
                    for (s = 0.0, k = l; k <= n; k++) s += mat[i - 1][k - 1] * v[k - 1][j - 1];
                    for (k = l; k <= n; k++) v[k - 1][j - 1] = (float)(v[k - 1][j - 1] + s * v[k - 1][i - 1]);
                }
            }
            for (j = l; j <= n; j++) v[i - 1][j - 1] = v[j - 1][i - 1] = 0.0;
        }
        v[i - 1][i - 1] = 1.0;
        g = rv1[i - 1];
        l = i;
    }
    for (i = (m < n ? m : n); i >= 1; i--) {
        l = i + 1;
        g = w[i - 1];
        for (j = l; j <= n; j++) mat[i - 1][j - 1] = 0.0;
        if (g) {
            g = 1.0 / g;
            for (j = l; j <= n; j++) {
                for (s = 0.0, k = l; k <= m; k++) s += mat[k - 1][i - 1] * mat[k - 1][j - 1];
                f = (s / mat[i - 1][i - 1]) * g;
                for (k = i; k <= m; k++) mat[k - 1][j - 1] = (float)(mat[k - 1][j - 1] + f * mat[k - 1][i - 1]);
            }
            for (j = i; j <= m; j++) mat[j - 1][i - 1] = (float)(mat[j - 1][i - 1] * g);
        } else
            for (j = i; j <= m; j++) mat[j - 1][i - 1] = 0.0;
        ++mat[i - 1][i - 1];
    }
    for (k = n; k >= 1; k--) {
        for (its = 1; its <= 30; its++) {
            flag = 1;
            for (l = k; l >= 1; l--) {
                nm = l - 1;
                if ((double)(fabs(rv1[l - 1]) + anorm) == anorm) {
                    flag = 0;
                    break;
                }
                if ((double)(fabsf(w[nm - 1]) + anorm) == anorm) break;
            }
            if (flag) {
                c = 0.0;
                s = 1.0;
                for (i = l; i <= k; i++) {
                    f = s * rv1[i - 1];
                    rv1[i - 1] = c * rv1[i - 1];
                    if ((double)(fabs(f) + anorm) == anorm) break;
                    g = w[i - 1];
                    h = SVD_PYTHAG(f, g);
                    w[i - 1] = (float)h;
                    h = 1.0 / h;
                    c = g * h;
                    s = -f * h;
                    for (j = 1; j <= m; j++) {
                        y = mat[j - 1][nm - 1];
                        z = mat[j - 1][i - 1];
                        mat[j - 1][nm - 1] = (float)(y * c + z * s);
                        mat[j - 1][i - 1] = (float)(z * c - y * s);
                    }
                }
            }
            z = w[k - 1];
            if (l == k) {
                if (z < 0.0) {
                    w[k - 1] = (float)(-z);
                    for (j = 1; j <= n; j++) v[j - 1][k - 1] = -v[j - 1][k - 1];
                }
                break;
            }
            x = w[l - 1];
            nm = k - 1;
            y = w[nm - 1];
            g = rv1[nm - 1];
            h = rv1[k - 1];
            f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y);
            g = SVD_PYTHAG(f, 1.0);
            f = ((x - z) * (x + z) + h * ((y / (f + SVD_SIGN(g, f))) - h)) / x;
            c = s = 1.0;
            for (j = l; j <= nm; j++) {
                i = j + 1;
                g = rv1[i - 1];
                y = w[i - 1];
                h = s * g;
                g = c * g;
                z = SVD_PYTHAG(f, h);
                rv1[j - 1] = z;
                c = f / z;
                s = h / z;
                f = x * c + g * s;
                g = g * c - x * s;
                h = y * s;
                y *= c;
                for (jj = 1; jj <= n; jj++) {
                    x = v[jj - 1][j - 1];
                    z = v[jj - 1][i - 1];
                    v[jj - 1][j - 1] = (float)(x * c + z * s);
                    v[jj - 1][i - 1] = (float)(z * c - x * s);
                }
                z = SVD_PYTHAG(f, h);
                w[j - 1] = (float)z;
                if (z) {
                    z = 1.0 / z;
                    c = f * z;
                    s = h * z;
                }
                f = c * g + s * y;
                x = c * y - s * g;
                for (jj = 1; jj <= m; jj++) {
                    y = mat[jj - 1][j - 1];
                    z = mat[jj - 1][i - 1];
                    mat[jj - 1][j - 1] = (float)(y * c + z * s);
                    mat[jj - 1][i - 1] = (float)(z * c - y * s);
                }
            }
            rv1[l - 1] = 0.0;
            rv1[k - 1] = f;
            w[k - 1] = (float)x;
        }
    }
}

/* R/src_common/SVD.h:15-31 SortEigenDecomp<float,3> */
void o3_sort_eig(float w[3], float v[3][3])
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

/* R/src_common/MultiScale.cpp:2541-2607 determineOrientation3D */
void o3_orientation(o3_feature *ft)
{
    float dx[PV], dy[PV], dz[PV];
    patch_edges(ft->data, dx, dy, dz);
    float mat[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const float r2 = (float)((PD / 2) * (PD / 2));
    for (int zz = 0; zz < PD; zz++)
        for (int yy = 0; yy < PD; yy++)
            for (int xx = 0; xx < PD; xx++) {
                float fz = (float)(zz - PD / 2), fy = (float)(yy - PD / 2), fx = (float)(xx - PD / 2);
                if (fz * fz + fy * fy + fx * fx < r2) {
                    int idx = (zz * PD + yy) * PD + xx;
                    float e[3] = {dx[idx], dy[idx], dz[idx]};
                    for (int i = 0; i < 3; i++)
                        for (int j = 0; j < 3; j++) mat[i][j] += e[i] * e[j];
                }
            }
    o3_svd3(mat, ft->eigs, ft->ori);
    o3_sort_eig(ft->eigs, ft->ori);
}

/* R/src_common/MultiScale.cpp:1092-1127 vec3D_norm_3d / vec3D_mag, :1571-1578 dot */
static void v3_norm(float *p)
{
    float ss = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if (ss > 0) {
        float div = (float)(1.0 / sqrtf(ss));
        p[0] *= div; p[1] *= div; p[2] *= div;
    } else {
        p[0] = 1; p[1] = 0; p[2] = 0;
    }
}
static float v3_mag(const float *p)
{
    float ss = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if (ss > 0) return sqrtf(ss);
    return 0;
}
static float v3_dot(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* R/src_common/LocationValue.cpp:28-56 lvSortHighLow: qsort whose comparator
 * returns 0 on ties; glibc 2.35 qsort is a merge sort (stable) for these
 * sizes, restated as a stable insertion sort by descending value. */
void o3_sort_high_low(o3_extremum *e, int n)
{
    for (int i = 1; i < n; i++) {
        o3_extremum t = e[i];
        int j = i - 1;
        while (j >= 0 && e[j].value < t.value) {
            e[j + 1] = e[j];
            j--;
        }
        e[j + 1] = t;
    }
}

/* R/src_common/MultiScale.cpp:1987-2121 regFindFEATUREIOPeaks without callback on an 11^3 grid */

/* Diagnostic (O3_TRACE_PEAKS=1): report orientation-histogram peaks near their keep threshold (0.8 of the
 * strongest primary, 0.5 of the strongest secondary: MultiScale.cpp:2889,2972-2985).  A frame whose ratio sits on the
 * threshold is where two builds of the same pipeline may legitimately differ by one record. */
static double o3_trace_band = 1e-3; /* O3_TRACE_PEAKS=<band>: half-width of the reported ratio band (default 1e-3) */
static int o3_trace_peaks(void)
{
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("O3_TRACE_PEAKS");
        on = e != NULL;
        if (e && atof(e) > 0 && atof(e) != 1.0) o3_trace_band = atof(e);
    }
    return on;
}

static int patch_peaks(const float *g, o3_extremum *out)
{
    int n = 0;
    for (int z = 1; z < PD - 1; z++)
        for (int y = 1; y < PD - 1; y++)
            for (int x = 1; x < PD - 1; x++) {
                int idx = (z * PD + y) * PD + x;
                float c = g[idx];
                int pk = 1;
                for (int dz = -1; dz <= 1 && pk; dz++)
                    for (int dy = -1; dy <= 1 && pk; dy++)
                        for (int dx = -1; dx <= 1 && pk; dx++) {
                            if (!dz && !dy && !dx) continue;
                            pk &= (g[idx + (dz * PD + dy) * PD + dx] < c);
                        }
                if (pk) {
                    out[n].x = x; out[n].y = y; out[n].z = z; out[n].value = c;
                    n++;
                }
            }
    return n;
}

/* Diagnostic companion of o3_trace_peaks(): cells of a blurred orientation histogram that fail the strict 26-neighbour
 * test of patch_peaks only because of neighbours within 1e-5 (relative) of their own value, i.e. peaks another build of
 * the same arithmetic (other rounding of the splat sums) may or may not see. */
static void o3_trace_near_ties(const float *g, const o3_feature *ft, int primary)
{
    for (int z = 1; z < PD - 1; z++)
        for (int y = 1; y < PD - 1; y++)
            for (int x = 1; x < PD - 1; x++) {
                const int idx = (z * PD + y) * PD + x;
                const float c = g[idx];
                if (!(c > 0)) continue;
                int ties = 0, above = 0;
                float worst = 0;
                for (int dz = -1; dz <= 1; dz++)
                    for (int dy = -1; dy <= 1; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            if (!dz && !dy && !dx) continue;
                            const float v = g[idx + (dz * PD + dy) * PD + dx];
                            if (v < c) continue;
                            if ((double)v - (double)c <= 1e-5 * (double)c) { ties++; if (v - c > worst) worst = v - c; }
                            else above++;
                        }
                if (ties > 0 && above == 0)
                    fprintf(stderr, "o3 peak-trace: kp (%.3f, %.3f, %.3f) primary %d: NEAR-TIE cell (%d,%d,%d) value %.9g, %d neighbour(s) "
                            "within 1e-5 relative (largest excess %.3g = %.2f ulp)\n", ft->x, ft->y, ft->z, primary, x, y, z, c, ties,
                            worst, (double)worst / (double)(nextafterf(c, 2 * c) - c));
            }
}

/* R/src_common/MultiScale.cpp:2722-3037 determineCanonicalOrientation3D.
 * ori_out receives up to max_ori 3x3 matrices (rows P1, P2, P3); the feature's
 * patch is zeroed as a side effect, as there (:2883). */
int o3_canonical_orientations(o3_feature *ft, float *ori_out, int max_ori)
{
    float dx[PV], dy[PV], dz[PV];
    float t0[PV], t2[PV];
    o3_extremum pk[PV], pk2[PV];
    float ori_data[PD * 3 + 3];
    float taps[O3_MAX_TAPS];
    const float blur_sigma = 0.5f; /* fBlurGradOriHist, MultiScale.cpp:37 */
    int ntaps = o3_gauss_taps(blur_sigma, 0.01f, taps);
    const float radius = (float)(PD / 2);
    const float r2 = (float)((PD / 2) * (PD / 2));

    memset(t0, 0, sizeof(t0));
    patch_edges(ft->data, dx, dy, dz);
    for (int zz = 0; zz < PD; zz++)
        for (int yy = 0; yy < PD; yy++)
            for (int xx = 0; xx < PD; xx++) {
                float fz = (float)(zz - PD / 2), fy = (float)(yy - PD / 2), fx = (float)(xx - PD / 2);
                if (fz * fz + fy * fy + fx * fx < r2) {
                    int idx = (zz * PD + yy) * PD + xx;
                    float e[3] = {dx[idx], dy[idx], dz[idx]};
                    float m2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
                    if (m2 == 0) continue;
                    float mag = sqrtf(m2);
                    float u[3];
                    for (int i = 0; i < 3; i++) u[i] = e[i] * radius / mag;
                    for (int i = 0; i < 3; i++) u[i] += radius;
                    splat(t0, PD, PD, PD, 1, (float)(u[0] + 0.5), (float)(u[1] + 0.5), (float)(u[2] + 0.5), 0, mag);
                }
            }
    o3_filter3d(t0, t2, PD, PD, PD, taps, ntaps);
    int npk = patch_peaks(t2, pk);
    if (o3_trace_peaks()) o3_trace_near_ties(t2, ft, -1);
    o3_sort_high_low(pk, npk);

    for (int i = 0; i < npk && i < PD && i < max_ori; i++) {
        float *oc = &ori_data[i * 3];
        o3_interp_point(t2, PD, PD, PD, pk[i].x, pk[i].y, pk[i].z, &oc[0], &oc[1], &oc[2]);
        oc[0] -= radius; oc[1] -= radius; oc[2] -= radius;
        v3_norm(oc);
    }

    int nret = 0;
    memset(ft->data, 0, sizeof(ft->data));
    for (int i = 0; i < npk && i < PD && nret < max_ori; i++) {
        if (o3_trace_peaks() && i > 0 && fabs((double)pk[i].value / (double)pk[0].value - 0.8) < o3_trace_band)
            fprintf(stderr, "o3 peak-trace: kp (%.3f, %.3f, %.3f) primary %d: value %.9g / max %.9g = %.9f -> %s\n", ft->x, ft->y,
                    ft->z, i, pk[i].value, pk[0].value, (double)pk[i].value / (double)pk[0].value,
                    pk[i].value < 0.8 * pk[0].value ? "rejected" : "kept");
        if (pk[i].value < 0.8 * pk[0].value) break;
        float p1[3], p2[3], p3[3];
        p1[0] = ori_data[i * 3]; p1[1] = ori_data[i * 3 + 1]; p1[2] = ori_data[i * 3 + 2];
        memset(t0, 0, sizeof(t0));
        for (int zz = 0; zz < PD; zz++)
            for (int yy = 0; yy < PD; yy++)
                for (int xx = 0; xx < PD; xx++) {
                    float fx = (float)(xx - PD / 2), fy = (float)(yy - PD / 2), fz = (float)(zz - PD / 2);
                    float lr2 = fz * fz + fy * fy + fx * fx;
                    if (lr2 < r2) {
                        int idx = (zz * PD + yy) * PD + xx;
                        float e[3] = {dx[idx], dy[idx], dz[idx]};
                        float mag = v3_mag(e);
                        if (mag == 0) continue;
                        float u[3] = {e[0], e[1], e[2]};
                        v3_norm(u);
                        float par = v3_dot(p1, u);
                        float pp[3];
                        pp[0] = u[0] - par * p1[0];
                        pp[1] = u[1] - par * p1[1];
                        pp[2] = u[2] - par * p1[2];
                        v3_norm(pp);
                        for (int k = 0; k < 3; k++) {
                            pp[k] *= radius;
                            pp[k] += radius;
                        }
                        splat(t0, PD, PD, PD, 1, (float)(pp[0] + 0.5), (float)(pp[1] + 0.5), (float)(pp[2] + 0.5), 0, mag);
                    }
                }
        o3_filter3d(t0, t2, PD, PD, PD, taps, ntaps);
        int npk2 = patch_peaks(t2, pk2);
        if (o3_trace_peaks()) o3_trace_near_ties(t2, ft, i);
        o3_sort_high_low(pk2, npk2);
        for (int j = 0; j < npk2 && nret < PD && nret < max_ori; j++) {
            if (o3_trace_peaks() && j > 0 && fabs((double)pk2[j].value / (double)pk2[0].value - 0.5) < o3_trace_band)
                fprintf(stderr, "o3 peak-trace: kp (%.3f, %.3f, %.3f) primary %d secondary %d: value %.9g / max %.9g = %.9f -> %s\n",
                        ft->x, ft->y, ft->z, i, j, pk2[j].value, pk2[0].value, (double)pk2[j].value / (double)pk2[0].value,
                        pk2[j].value < 0.5f * pk2[0].value ? "rejected" : "kept");
            if (pk2[j].value < 0.5f * pk2[0].value) break; /* fHist2ndPeakThreshold, MultiScale.cpp:40 */
            o3_interp_point(t2, PD, PD, PD, pk2[j].x, pk2[j].y, pk2[j].z, &p2[0], &p2[1], &p2[2]);
            p2[0] -= radius; p2[1] -= radius; p2[2] -= radius;
            v3_norm(p2);
            float par = v3_dot(p1, p2);
            p2[0] = p2[0] - par * p1[0];
            p2[1] = p2[1] - par * p1[1];
            p2[2] = p2[2] - par * p1[2];
            v3_norm(p2);
            /* MultiScale.cpp:3039-3049 vec3D_cross_3d */
            p3[0] = p1[1] * p2[2] - p1[2] * p2[1];
            p3[1] = -p1[0] * p2[2] + p1[2] * p2[0];
            p3[2] = p1[0] * p2[1] - p1[1] * p2[0];
            float *m = ori_out + 9 * nret;
            for (int iv = 0; iv < 3; iv++) {
                m[0 * 3 + iv] = p1[iv];
                m[1 * 3 + iv] = p2[iv];
                m[2 * 3 + iv] = p3[iv];
            }
            nret++;
        }
    }
    return nret;
}

/* ------------------------------------------------------------------------ */
/* Descriptors                                                              */
/* ------------------------------------------------------------------------ */

/* R/src_common/MultiScale.cpp:1580-1611 msNormalizeDataPositive */
static void normalize_positive(float *v, int n)
{
    float mn = 100000;
    for (int i = 0; i < n; i++)
        if (v[i] < mn) mn = v[i];
    float ss = 0;
    for (int i = 0; i < n; i++) {
        v[i] -= mn;
        ss += v[i] * v[i];
    }
    float div = 1.0f / sqrtf(ss);
    for (int i = 0; i < n; i++) v[i] *= div;
}

/* R/src_common/MultiScale.cpp:583-710 msResampleFeaturesGradientOrientationHistogram */
void o3_desc_sift(o3_feature *ft)
{
    float dx[PV], dy[PV], dz[PV];
    patch_edges(ft->data, dx, dy, dz);
    static const float oa[8][3] = {{1, 1, 1},  {1, 1, -1},  {1, -1, 1},  {1, -1, -1},
                                   {-1, 1, 1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};
    const float bin = PD / (float)2;
    float coord[PD];
    for (int c = 0; c < PD; c++) {
        float v = (int)(c / bin) + 0.5f;
        if ((int)((c + 0) / bin) != (int)((c + 1) / bin)) {
            float p0 = ((c + 0) / bin);
            float p1 = ((c + 1) / bin);
            v = (p0 + p1) / 2.0f;
        }
        coord[c] = v;
    }
    for (int zz = 0; zz < PD; zz++)
        for (int yy = 0; yy < PD; yy++)
            for (int xx = 0; xx < PD; xx++) {
                int idx = (zz * PD + yy) * PD + xx;
                float e[3] = {dx[idx], dy[idx], dz[idx]};
                float mag = v3_mag(e);
                if (mag > 0) {
                    v3_norm(e);
                    int best = 0;
                    float bd = v3_dot(oa[0], e);
                    for (int k = 1; k < 8; k++) {
                        float d = v3_dot(oa[k], e);
                        if (d > bd) {
                            bd = d;
                            best = k;
                        }
                    }
                    splat(ft->pc, 2, 2, 2, 8, coord[xx], coord[yy], coord[zz], best, mag);
                }
            }
    normalize_positive(ft->pc, O3_DESC_LEN);
}

/* Pair tables of msGenerateBRIEFindex method 2 (data constants,
 * R/src_common/MultiScale.cpp:805-807): 64 (x,y,z) triples each. */
static const unsigned char brief_x[192] = {
    5,4,4,4,4,2,6,5,5,4,4,4,3,8,5,5,6,3,5,5,5,5,6,5,4,6,6,6,3,4,4,4,5,3,4,5,4,5,5,4,2,7,7,5,3,5,4,5,3,5,7,3,5,5,2,3,5,5,6,6,4,6,5,4,
    4,6,5,3,5,6,4,3,6,4,4,5,3,3,3,6,6,5,2,4,4,6,3,6,3,2,3,5,4,5,3,4,3,6,5,4,3,6,4,5,2,4,3,7,2,3,6,5,2,6,3,3,5,6,3,6,3,5,3,6,5,7,4,2,
    5,5,5,2,5,7,4,2,5,3,4,3,3,7,4,4,7,6,4,4,2,8,7,6,5,4,7,3,6,6,5,2,4,5,3,2,5,5,1,6,3,6,3,6,2,5,4,4,7,2,6,3,2,2,4,3,3,2,3,4,2,5,6,7};
static const unsigned char brief_y[192] = {
    6,5,3,4,5,3,7,4,6,4,3,2,4,7,5,3,5,1,5,4,7,6,8,4,4,5,6,5,2,5,4,6,4,0,4,3,3,4,4,2,1,7,8,6,4,4,1,6,1,3,7,2,3,3,1,3,6,1,6,6,4,7,6,4,
    3,5,4,2,3,6,4,5,6,3,3,5,1,3,1,6,7,4,1,4,3,5,2,4,2,1,2,5,4,5,2,3,3,3,3,4,2,6,3,4,3,3,3,6,1,2,5,4,2,4,1,4,6,7,3,6,2,4,3,6,5,6,4,0,
    6,6,5,1,4,7,2,1,5,3,4,2,2,7,3,3,6,4,2,4,1,9,7,7,5,2,7,1,7,5,5,1,5,4,1,3,3,4,0,5,1,6,3,5,3,2,3,3,7,2,5,1,1,0,4,1,3,1,0,3,1,6,5,9};

void o3_brief_tables(int *x_idx, int *y_idx)
{
    for (int i = 0; i < 64; i++) {
        x_idx[i] = brief_x[3 * i] + brief_x[3 * i + 1] * PD + brief_x[3 * i + 2] * PD * PD;
        y_idx[i] = brief_y[3 * i] + brief_y[3 * i + 1] * PD + brief_y[3 * i + 2] * PD * PD;
    }
}

/* R/src_common/MultiScale.cpp:989-1049 msResampleFeaturesBRIEF with the three
 * alternatives of :1037-1045; :1051-1056 euclidean_distance_3d.  The patch
 * blur uses the CPU pass order x,y,z. */
void o3_desc_brief(o3_feature *ft, int mode)
{
    float bl[PV];
    o3_blur(ft->data, bl, PD, PD, PD, 0.95, 0.01);
    for (int i = 0; i < O3_DESC_LEN; i++) {
        int x1 = brief_x[3 * i], y1 = brief_x[3 * i + 1], z1 = brief_x[3 * i + 2];
        int x2 = brief_y[3 * i], y2 = brief_y[3 * i + 1], z2 = brief_y[3 * i + 2];
        float d = bl[x1 + y1 * PD + z1 * PD * PD] - bl[x2 + y2 * PD + z2 * PD * PD];
        if (mode == O3_DESC_BRIEF) {
            ft->pc[i] = d < 0;
        } else if (mode == O3_DESC_RRIEF) {
            ft->pc[i] = d;
        } else {
            float fdx = x1 - x2, fdy = y1 - y2, fdz = z1 - z2;
            int dist = (int)sqrtf(fdx * fdx + fdy * fdy + fdz * fdz);
            ft->pc[i] = d / dist;
        }
    }
}

/* R/src_common/MultiScale.cpp:207-233 NormalizeDataRankedPCs with the total
 * order of :3148-3176 (value ascending, ties by lower index) */
void o3_rank(float *pc)
{
    int idx[O3_DESC_LEN];
    float val[O3_DESC_LEN];
    for (int i = 0; i < O3_DESC_LEN; i++) {
        idx[i] = i;
        val[i] = pc[i];
    }
    for (int i = 1; i < O3_DESC_LEN; i++) { /* stable insertion sort == that total order */
        int t = idx[i];
        float v = val[t];
        int j = i - 1;
        while (j >= 0 && (val[idx[j]] > v)) {
            idx[j + 1] = idx[j];
            j--;
        }
        idx[j + 1] = t;
    }
    for (int k = 0; k < O3_DESC_LEN; k++) pc[idx[k]] = (float)k;
}

/* ------------------------------------------------------------------------ */
/* Per-keypoint driver                                                      */
/* ------------------------------------------------------------------------ */
typedef struct {
    o3_feature *v;
    int64_t n, cap;
} featvec;

static void fv_push(featvec *fv, const o3_feature *f)
{
    if (fv->n == fv->cap) {
        fv->cap = fv->cap ? fv->cap * 2 : 256;
        fv->v = (o3_feature *)realloc(fv->v, sizeof(o3_feature) * (size_t)fv->cap);
    }
    fv->v[fv->n++] = *f;
}

/* R/src_common/MultiScale.cpp:1705-1862 generateFeature3D.  Returns 1 if the
 * keypoint survived (bounds + eigen test), else 0. */
static int generate_feature(o3_feature *ft, const float *img, int64_t X, int64_t Y, int64_t Z, featvec *fv, float eig_thres)
{
    float patch[PV];
    memset(ft->ori, 0, sizeof(ft->ori));
    ft->ori[0][0] = 1; ft->ori[1][1] = 1; ft->ori[2][2] = 1;
    if (o3_sample_patch(ft, img, X, Y, Z, patch) != 0) return 0;
    memcpy(ft->data, patch, sizeof(patch));
    o3_normalize_patch(ft->data);
    o3_orientation(ft);
    float es = ft->eigs[0] + ft->eigs[1] + ft->eigs[2];
    float ep = ft->eigs[0] * ft->eigs[1] * ft->eigs[2];
    float esp = es * es * es;
    if (esp < eig_thres * ep || eig_thres < 0) {
    } else {
        return 0;
    }
    ft->info &= ~O3_INFO_REORIENT;
    fv_push(fv, ft);

    float oris[30 * 9];
    int nori = o3_canonical_orientations(ft, oris, 30);
    for (int io = 0; io < nori; io++) {
        const float *o = oris + 9 * io;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) ft->ori[i][j] = o[i * 3 + j];
        if (o3_sample_patch(ft, img, X, Y, Z, patch) != 0) continue;
        memcpy(ft->data, patch, sizeof(patch));
        ft->info |= O3_INFO_REORIENT;
        fv_push(fv, ft);
    }
    return 1;
}

/* R/src_common/MultiScale.cpp:1326-1424 generateFeatures3D_efficient */
static int64_t generate_features(o3_feature *ft, const o3_extremum *mins, const float *minH, const float *minL, int64_t nmin,
                                 const o3_extremum *maxs, const float *maxH, const float *maxL, int64_t nmax,
                                 const float *C, float sH, float sC, float sL, const float *img, int64_t X, int64_t Y,
                                 int64_t Z, featvec *fv, float eig_thres)
{
    int64_t kept = 0;
#ifdef _OPENMP
    if (nmin + nmax >= 64) {
        /* Keypoints are independent of each other (every field of the shared record is rewritten per keypoint), so each
         * thread works on a record of its own and collects a keypoint's records in a list of that keypoint; the lists are
         * appended in candidate order afterwards: the output is the serial output. */
        const int64_t total = nmin + nmax;
        featvec *lv = (featvec *)calloc((size_t)total, sizeof(featvec));
#pragma omp parallel reduction(+ : kept)
        {
            o3_feature *f = (o3_feature *)malloc(sizeof(o3_feature));
            *f = *ft;
#pragma omp for schedule(dynamic, 8)
            for (int64_t q = 0; q < total; q++) {
                const int pass = q >= nmin;
                const int64_t i = pass ? q - nmin : q;
                const o3_extremum *e = pass == 0 ? mins : maxs;
                const float *eh = pass == 0 ? minH : maxH;
                const float *el = pass == 0 ? minL : maxL;
                o3_interp_point(C, X, Y, Z, e[i].x, e[i].y, e[i].z, &f->x, &f->y, &f->z);
                float cv = C[((int64_t)e[i].z * Y + e[i].y) * X + e[i].x];
                f->scale = (float)(2 * o3_interp_quadratic(sH, sC, sL, eh[i], cv, el[i]));
                f->x += 0.5f; f->y += 0.5f; f->z += 0.5f;
                if (pass == 0) f->info &= ~O3_INFO_MIN0MAX1;
                else f->info |= O3_INFO_MIN0MAX1;
                kept += generate_feature(f, img, X, Y, Z, &lv[q], eig_thres);
            }
            free(f);
        }
        int64_t add = 0;
        for (int64_t q = 0; q < total; q++) add += lv[q].n;
        if (fv->n + add > fv->cap) {
            fv->cap = fv->n + add + add / 4 + 256;
            fv->v = (o3_feature *)realloc(fv->v, sizeof(o3_feature) * (size_t)fv->cap);
        }
        int64_t *at = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total + 1));
        at[0] = fv->n;
        for (int64_t q = 0; q < total; q++) at[q + 1] = at[q] + lv[q].n;
#pragma omp parallel for schedule(static)
        for (int64_t q = 0; q < total; q++) {
            if (lv[q].n) memcpy(fv->v + at[q], lv[q].v, sizeof(o3_feature) * (size_t)lv[q].n);
            free(lv[q].v);
        }
        fv->n = at[total];
        free(at);
        free(lv);
        return kept;
    }
#endif
    for (int pass = 0; pass < 2; pass++) {
        const o3_extremum *e = pass == 0 ? mins : maxs;
        const float *eh = pass == 0 ? minH : maxH;
        const float *el = pass == 0 ? minL : maxL;
        int64_t n = pass == 0 ? nmin : nmax;
        for (int64_t i = 0; i < n; i++) {
            o3_interp_point(C, X, Y, Z, e[i].x, e[i].y, e[i].z, &ft->x, &ft->y, &ft->z);
            float cv = C[((int64_t)e[i].z * Y + e[i].y) * X + e[i].x];
            ft->scale = (float)(2 * o3_interp_quadratic(sH, sC, sL, eh[i], cv, el[i]));
            ft->x += 0.5f; ft->y += 0.5f; ft->z += 0.5f;
            if (pass == 0) ft->info &= ~O3_INFO_MIN0MAX1;
            else ft->info |= O3_INFO_MIN0MAX1;
            kept += generate_feature(ft, img, X, Y, Z, fv, eig_thres);
        }
    }
    return kept;
}

/* ------------------------------------------------------------------------ */
/* Pyramid driver: R/src_common/MultiScale.cpp:236-570                       */
/* msGeneratePyramidDOG3D_efficient.  Buffers are not recycled as there      */
/* (values are what matters); the schedule is the same: per octave 5 blurs,  */
/* DoG k = L_k - L_{k+1}, detection in DoG 1..3 against DoG k-1, validation  */
/* against DoG k+1, keypoints sampled from L_k, octave seeded by the         */
/* subsampled L_3.                                                          */
/* ------------------------------------------------------------------------ */
typedef struct {
    featvec *fv;           /* NULL: no per-keypoint work */
    o3_candidate *cand;    /* optional candidate log */
    int64_t ncand, capcand;
    o3_stats st;
} pyr_sink;

static void cand_push(pyr_sink *s, int oct, int lvl, int is_max, const o3_extremum *e, float h, float l)
{
    if (s->ncand == s->capcand) {
        s->capcand = s->capcand ? 2 * s->capcand : 1024;
        s->cand = (o3_candidate *)realloc(s->cand, sizeof(o3_candidate) * (size_t)s->capcand);
    }
    o3_candidate *c = &s->cand[s->ncand++];
    c->octave = oct; c->level = lvl; c->is_max = is_max;
    c->x = e->x; c->y = e->y; c->z = e->z;
    c->value = e->value; c->h_value = h; c->l_value = l;
}

static int run_pyramid(const float *vol, int64_t X, int64_t Y, int64_t Z, float init_scale, float eig_thres, pyr_sink *sink,
                       int want_cand)
{
    int64_t N = X * Y * Z;
    float *L[6], *D[5];
    for (int i = 0; i < 6; i++) L[i] = (float *)malloc(sizeof(float) * (size_t)N);
    for (int i = 0; i < 5; i++) D[i] = (float *)malloc(sizeof(float) * (size_t)N);
    float *half = (float *)malloc(sizeof(float) * (size_t)((X / 2) * (Y / 2) * (Z / 2) + 1));
    int64_t cap = N / 8 + 1024;
    o3_extremum *mins = (o3_extremum *)malloc(sizeof(o3_extremum) * (size_t)cap);
    o3_extremum *maxs = (o3_extremum *)malloc(sizeof(o3_extremum) * (size_t)cap);
    float *minH = (float *)malloc(sizeof(float) * (size_t)cap), *minL = (float *)malloc(sizeof(float) * (size_t)cap);
    float *maxH = (float *)malloc(sizeof(float) * (size_t)cap), *maxL = (float *)malloc(sizeof(float) * (size_t)cap);
    o3_feature *ft = (o3_feature *)calloc(1, sizeof(o3_feature)); /* one record reused across keypoints, as there */

    float sigma_init = 0.5f;
    if (init_scale > 0) sigma_init /= init_scale;
    float sigma = 1.6f;
    const float factor = (float)pow(2.0, 1.0 / (double)3);
    float extra = sqrtf(sigma * sigma - sigma_init * sigma_init);
    double t0 = now_s();
    o3_blur(vol, L[0], X, Y, Z, extra, 0.01f);
    sink->st.t_blur += now_s() - t0;

    float fscale = 1;
    float sig[7];
    for (int oct = 0;; oct++) {
        sigma = 1.6f;
        sig[0] = sigma;
        if (X <= 2 || Y <= 2 || Z <= 2) break;
        N = X * Y * Z;
        int64_t first = sink->fv ? sink->fv->n : 0;
        int64_t nmin = 0, nmax = 0;
        for (int j = 1; j < 6; j++) {
            float ex = sigma * sqrtf(factor * factor - 1.0f);
            t0 = now_s();
            o3_blur(L[j - 1], L[j], X, Y, Z, ex, 0.01f);
            sink->st.t_blur += now_s() - t0;
            if (j == 1) {
                t0 = now_s();
                o3_dog(L[0], L[1], D[0], N);
                sink->st.t_dog += now_s() - t0;
            } else {
                if (j == 3) {
                    t0 = now_s();
                    o3_subsample(L[3], X, Y, Z, half);
                    sink->st.t_subsample += now_s() - t0;
                }
                if (j >= 3) {
                    /* validate the candidates found in DoG j-2 against L[j-1] - L[j] */
                    t0 = now_s();
                    int64_t top = 0;
                    for (int64_t k = 0; k < nmax; k++)
                        if (o3_validate_peak(&maxs[k], L[j - 1], L[j], X, Y, Z)) {
                            int64_t idx = ((int64_t)maxs[k].z * Y + maxs[k].y) * X + maxs[k].x;
                            maxH[top] = maxH[k];
                            maxL[top] = L[j - 1][idx] - L[j][idx];
                            maxs[top] = maxs[k];
                            top++;
                        }
                    nmax = top;
                    top = 0;
                    for (int64_t k = 0; k < nmin; k++)
                        if (o3_validate_valley(&mins[k], L[j - 1], L[j], X, Y, Z)) {
                            int64_t idx = ((int64_t)mins[k].z * Y + mins[k].y) * X + mins[k].x;
                            minH[top] = minH[k];
                            minL[top] = L[j - 1][idx] - L[j][idx];
                            mins[top] = mins[k];
                            top++;
                        }
                    nmin = top;
                    sink->st.t_detect += now_s() - t0;
                    sink->st.n_extrema += nmin + nmax;
                    if (want_cand) {
                        for (int64_t k = 0; k < nmin; k++) cand_push(sink, oct, j - 2, 0, &mins[k], minH[k], minL[k]);
                        for (int64_t k = 0; k < nmax; k++) cand_push(sink, oct, j - 2, 1, &maxs[k], maxH[k], maxL[k]);
                    }
                    if (sink->fv) {
                        t0 = now_s();
                        sink->st.n_keypoints +=
                            generate_features(ft, mins, minH, minL, nmin, maxs, maxH, maxL, nmax, D[j - 2], sig[j - 3],
                                              sig[j - 2], sig[j - 1], L[j - 2], X, Y, Z, sink->fv, eig_thres);
                        sink->st.t_features += now_s() - t0;
                    }
                }
                if (j < 5) {
                    t0 = now_s();
                    o3_dog(L[j - 1], L[j], D[j - 1], N);
                    sink->st.t_dog += now_s() - t0;
                    t0 = now_s();
                    o3_detect(D[j - 2], D[j - 1], X, Y, Z, mins, cap, &nmin, maxs, cap, &nmax);
                    for (int64_t k = 0; k < nmax; k++)
                        maxH[k] = D[j - 2][((int64_t)maxs[k].z * Y + maxs[k].y) * X + maxs[k].x];
                    for (int64_t k = 0; k < nmin; k++)
                        minH[k] = D[j - 2][((int64_t)mins[k].z * Y + mins[k].y) * X + mins[k].x];
                    sink->st.t_detect += now_s() - t0;
                }
            }
            sigma *= factor;
            sig[j] = sigma;
        }
        /* octave -> image space, MultiScale.cpp:531-543 */
        if (sink->fv) {
            float fac = fscale, add = 0;
            for (int64_t i = first; i < sink->fv->n; i++) {
                o3_feature *f = &sink->fv->v[i];
                f->scale *= fac;
                f->x = f->x * fac + add;
                f->y = f->y * fac + add;
                f->z = f->z * fac + add;
            }
        }
        fscale *= 2.0f;
        X /= 2; Y /= 2; Z /= 2;
        memcpy(L[0], half, sizeof(float) * (size_t)(X * Y * Z));
        sink->st.n_octaves++;
    }
    free(ft);
    free(maxL); free(maxH); free(minL); free(minH);
    free(maxs); free(mins);
    free(half);
    for (int i = 0; i < 5; i++) free(D[i]);
    for (int i = 0; i < 6; i++) free(L[i]);
    return 1;
}

int o3_pyramid_features(const float *vol, int64_t X, int64_t Y, int64_t Z, float init_scale, float eig_thres,
                        o3_feature **out, int64_t *n_out, o3_stats *stats)
{
    featvec fv = {0, 0, 0};
    pyr_sink s;
    memset(&s, 0, sizeof(s));
    s.fv = &fv;
    run_pyramid(vol, X, Y, Z, init_scale, eig_thres, &s, 0);
    *out = fv.v;
    *n_out = fv.n;
    if (stats) *stats = s.st;
    return 1;
}

int o3_pyramid_candidates(const float *vol, int64_t X, int64_t Y, int64_t Z, float init_scale, o3_candidate **out,
                          int64_t *n_out)
{
    pyr_sink s;
    memset(&s, 0, sizeof(s));
    run_pyramid(vol, X, Y, Z, init_scale, 140.0f, &s, 1);
    *out = s.cand;
    *n_out = s.ncand;
    return 1;
}

int o3_octave_levels(const float *g0, int64_t X, int64_t Y, int64_t Z, float *G, float *D)
{
    const int64_t N = X * Y * Z;
    float sigma = 1.6f;
    const float factor = (float)pow(2.0, 1.0 / (double)3);
    memcpy(G, g0, sizeof(float) * (size_t)N);
    for (int j = 1; j < 6; j++) {
        float ex = sigma * sqrtf(factor * factor - 1.0f);
        o3_blur(G + (j - 1) * N, G + j * N, X, Y, Z, ex, 0.01f);
        o3_dog(G + (j - 1) * N, G + j * N, D + (j - 1) * N, N);
        sigma *= factor;
    }
    return 1;
}

/* featExtract.cpp:474-505 descriptor loop + size factor */
int o3_extract(const float *vol, int64_t X, int64_t Y, int64_t Z, float init_scale, int desc_mode, float eig_thres,
               float size_factor, o3_record **out, int64_t *n_out, o3_stats *stats)
{
    o3_feature *fs = 0;
    int64_t n = 0;
    o3_stats st;
    o3_pyramid_features(vol, X, Y, Z, init_scale, eig_thres, &fs, &n, &st);
    o3_record *r = (o3_record *)malloc(sizeof(o3_record) * (size_t)(n ? n : 1));
    double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int64_t i = 0; i < n; i++) {
        o3_feature *f = &fs[i];
        o3_normalize_patch(f->data);
        if (desc_mode == O3_DESC_SIFT) o3_desc_sift(f);
        else o3_desc_brief(f, desc_mode);
        o3_rank(f->pc);
        f->x *= size_factor; f->y *= size_factor; f->z *= size_factor; f->scale *= size_factor;
        r[i].x = f->x; r[i].y = f->y; r[i].z = f->z; r[i].scale = f->scale;
        memcpy(r[i].ori, f->ori, sizeof(float) * 9);
        memcpy(r[i].eigs, f->eigs, sizeof(float) * 3);
        r[i].info = f->info;
        memcpy(r[i].desc, f->pc, sizeof(float) * O3_DESC_LEN);
    }
    st.t_desc = now_s() - t0;
    free(fs);
    *out = r;
    *n_out = n;
    if (stats) *stats = st;
    return 1;
}

/* R/src_common/MultiScale.h:386-474 msFeature3DVectorOutputText */
int o3_write_key(const char *path, const o3_record *recs, int64_t n, float eig_thres, int n_comments, const char **comments)
{
    FILE *f = fopen(path, "wt");
    if (!f) return -1;
    int count = 0;
    for (int64_t i = 0; i < n; i++) {
        float es = recs[i].eigs[0] + recs[i].eigs[1] + recs[i].eigs[2];
        float ep = recs[i].eigs[0] * recs[i].eigs[1] * recs[i].eigs[2];
        float esp = es * es * es;
        if (esp < eig_thres * ep || eig_thres < 0) count++;
    }
    fprintf(f, "# featExtract %s\n", "1.1");
    for (int i = 0; i < n_comments; i++) fprintf(f, "# %s\n", comments[i]);
    fprintf(f, "Features: %d\n", count);
    fprintf(f, "Scale-space location[x y z scale] orientation[o11 o12 o13 o21 o22 o23 o31 o32 o32] 2nd moment "
               "eigenvalues[e1 e2 e3] info flag[i1] descriptor[d1 .. d64]\n");
    for (int64_t i = 0; i < n; i++) {
        const o3_record *r = &recs[i];
        float es = r->eigs[0] + r->eigs[1] + r->eigs[2];
        float ep = r->eigs[0] * r->eigs[1] * r->eigs[2];
        float esp = es * es * es;
        if (esp < eig_thres * ep || eig_thres < 0) {
        } else
            continue;
        fprintf(f, "%f\t%f\t%f\t%f\t", r->x, r->y, r->z, r->scale);
        for (int j = 0; j < 9; j++) fprintf(f, "%f\t", r->ori[j]);
        for (int j = 0; j < 3; j++) fprintf(f, "%f\t", r->eigs[j]);
        fprintf(f, "%d\t", r->info);
        for (int j = 0; j < O3_DESC_LEN; j++) fprintf(f, "%i\t", (char)(r->desc[j]));
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}
