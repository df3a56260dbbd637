/* tools/key_writer_bench.c -- the .key text writer alone on 185 421 records (a 512^3 extraction's worth, 63 MB of text): serial
 * (OMP_NUM_THREADS=1), parallel by positional writes (mode 0), parallel into the mapped file (mode 1).
 *   gcc -O2 -std=c11 -D_POSIX_C_SOURCE=200809L -fopenmp -Iinclude -I3d_sift_cuda_amd/csrc tools/key_writer_bench.c 3d_sift_cuda_amd/csrc/keyfile.c -o /tmp/kwb -lm
 *   /tmp/kwb out.key [mode=0|1] */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "keyfile.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    const int64_t n = 185421;
    sift3d_feature *r = calloc((size_t)n, sizeof *r);
    unsigned s = 1;
    for (int64_t i = 0; i < n; i++) {
        float *f = (float *)&r[i];
        for (int j = 0; j < 16; j++) { s = s * 1664525u + 1013904223u; f[j] = (s >> 8) / 16777216.0f * 500; }
        r[i].info = 16;
        for (int j = 0; j < 64; j++) r[i].desc[j] = (float)((j * 7 + i) % 64);
    }
    if (argc > 2) sift3d_write_key_mode(atoi(argv[2]));
    const char *cm[1] = {"a"};
    for (int k = 0; k < 6; k++) {
        const double t = now();
        if (sift3d_write_key(argv[1], r, n, -1.0f, 1, cm) != 0) return 1;
        printf("%.4f ", now() - t);
    }
    printf("\n");
    return 0;
}
