/* synth.c -- see synth.h */
#include "synth.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* the blobs of the generator, in generation order, accumulated into planes [za, zb) only; vol holds the planes from zbase on */
static void blobs_into_planes(float *vol, int64_t X, int64_t Y, int64_t Z, uint32_t seed, int64_t za, int64_t zb, int64_t zbase)
{
    const int64_t N = X * Y * Z;
    uint32_t s = seed;
#define NEXT_U() (s = s * 1664525u + 1013904223u, (float)(s >> 8) / 16777216.0f)
    const int64_t nblobs = N / 2048;
    for (int64_t b = 0; b < nblobs; b++) {
        float cx = NEXT_U() * (float)X;
        float cy = NEXT_U() * (float)Y;
        float cz = NEXT_U() * (float)Z;
        float sg = 1.5f + 4.0f * NEXT_U();
        float amp = 200.0f * (NEXT_U() - 0.3f);
        int r = (int)(3.0f * sg) + 1;
        int64_t x0 = (int64_t)cx - r, x1 = (int64_t)cx + r;
        int64_t y0 = (int64_t)cy - r, y1 = (int64_t)cy + r;
        int64_t z0 = (int64_t)cz - r, z1 = (int64_t)cz + r;
        if (x0 < 0) x0 = 0;
        if (y0 < 0) y0 = 0;
        if (z0 < 0) z0 = 0;
        if (x1 > X - 1) x1 = X - 1;
        if (y1 > Y - 1) y1 = Y - 1;
        if (z1 > Z - 1) z1 = Z - 1;
        if (z0 < za) z0 = za;
        if (z1 > zb - 1) z1 = zb - 1;
        for (int64_t z = z0; z <= z1; z++) {
            float dz = (float)z - cz;
            for (int64_t y = y0; y <= y1; y++) {
                float dy = (float)y - cy;
                float *row = vol + ((z - zbase) * Y + y) * X;
                for (int64_t x = x0; x <= x1; x++) {
                    float dx = (float)x - cx;
                    float d2 = dx * dx + dy * dy + dz * dz;
                    row[x] += amp * expf(-d2 / (2.0f * sg * sg));
                }
            }
        }
    }
#undef NEXT_U
}

void sift3d_synth_blobs(float *vol, int64_t X, int64_t Y, int64_t Z, uint32_t seed)
{
    sift3d_synth_blobs_slices(vol, X, Y, Z, seed, 0, Z);
}

void sift3d_synth_blobs_slices(float *out, int64_t X, int64_t Y, int64_t Z, uint32_t seed, int64_t z0, int64_t z1)
{
    if (z0 < 0) z0 = 0;
    if (z1 > Z) z1 = Z;
    if (z1 <= z0) return;
    const int64_t W = z1 - z0;
    /* A voxel's value is the float sum of its blobs in generation order.  Each thread walks the whole blob sequence (five
     * draws a blob) and adds only into its own range of planes, so every voxel still sees its blobs in that order: the
     * same bits for any thread count (a 2048 x 2048 x 1024 volume takes minutes on one core). */
#ifdef _OPENMP
    int nt = omp_get_max_threads();
    if (nt > 32) nt = 32; /* (a box of the pool shows 256 cores to a 16-CPU share; every thread walks the whole blob sequence) */
    if (nt > W) nt = (int)W;
    if (X * Y * W < (1ll << 22) || nt < 1) nt = 1;
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < nt; t++) {
        const int64_t za = z0 + W * t / nt, zb = z0 + W * (t + 1) / nt;
        memset(out + (za - z0) * Y * X, 0, sizeof(float) * (size_t)((zb - za) * Y * X));
        blobs_into_planes(out, X, Y, Z, seed, za, zb, z0);
    }
#else
    memset(out, 0, sizeof(float) * (size_t)(X * Y * W));
    blobs_into_planes(out, X, Y, Z, seed, z0, z1, z0);
#endif
}
