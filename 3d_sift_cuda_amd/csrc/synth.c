/* synth.c -- see synth.h */
#include "synth.h"

#include <math.h>
#include <string.h>

void sift3d_synth_blobs(float *vol, int64_t X, int64_t Y, int64_t Z, uint32_t seed)
{
    const int64_t N = X * Y * Z;
    memset(vol, 0, sizeof(float) * (size_t)N);
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
        for (int64_t z = z0; z <= z1; z++) {
            float dz = (float)z - cz;
            for (int64_t y = y0; y <= y1; y++) {
                float dy = (float)y - cy;
                float *row = vol + (z * Y + y) * X;
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
