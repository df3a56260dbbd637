/*
 * host_asan_driver.c -- test infrastructure: runs the product's plain-C host code (nifti_min.c, world.c, keyfile.c,
 * synth.c) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (the GPU pool has no sanitizer builds).
 * Built by `make -C 3d_sift_cuda_amd/csrc asan`; tests/test_abi_and_host.py feeds it the NIfTI files it generates.
 *
 *   host_asan_driver read <file>...      nifti_min_read every file (failures are fine, crashes are not), resample the
 *                                         readable ones to isotropic voxels (-w), print "<file> rc dims"
 *   host_asan_driver keys <dir>          write / read back text and binary .key files of synthetic records, apply a
 *                                         world transform, compare; image.pgm of a slice, of a constant slice
 *   host_asan_driver votes               the matcher's host side (match_votes.c, round 4): filters, descriptor bytes incl.
 *                                         values that are not bytes, votes on well-formed neighbour lists and on lists
 *                                         with indices / labels out of range (must be refused, not read or written through)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "keyfile.h"
#include "match.h"
#include "nifti_min.h"
#include "sift3d.h"
#include "synth.h"
#include "world.h"

int main(int argc, char **argv)
{
    if (argc >= 3 && strcmp(argv[1], "read") == 0) {
        for (int i = 2; i < argc; i++) {
            nifti_min_image img;
            int rc = nifti_min_read(argv[i], &img);
            printf("%s rc=%d", argv[i], rc);
            if (rc == 0) {
                double s = 0;
                const size_t n = (size_t)img.nx * img.ny * img.nz * img.nt;
                for (size_t k = 0; k < n; k++) s += img.data[k];
                printf(" dims=%dx%dx%dx%d sum=%.6g", img.nx, img.ny, img.nz, img.nt, s);
                if (img.nt == 1 && img.nz > 1) {
                    int rw = sift3d_world_make_isotropic(&img);
                    printf(" iso rc=%d dims=%dx%dx%d", rw, img.nx, img.ny, img.nz);
                }
            }
            printf("\n");
            nifti_min_free(&img);
        }
        return 0;
    }
    if (argc == 3 && strcmp(argv[1], "keys") == 0) {
        enum { N = 257 };
        sift3d_feature *f = (sift3d_feature *)calloc(N, sizeof(*f));
        float *v = (float *)malloc(sizeof(float) * 16 * 16 * 16);
        sift3d_synth_blobs(v, 16, 16, 16, 7u);
        for (int i = 0; i < N; i++) {
            f[i].x = v[i] + (float)i; f[i].y = v[i + 300] * 3.0f; f[i].z = -v[i + 700]; f[i].scale = 1.0f + (float)(i % 7);
            for (int k = 0; k < 9; k++) f[i].ori[k] = (k % 4 == 0) ? 1.0f : 0.0f;
            f[i].eigs[0] = 3.0f; f[i].eigs[1] = 2.0f; f[i].eigs[2] = 1.5f; /* passes the eigenvalue test the writer re-applies */
            f[i].info = (i & 1) ? 0x30u : 0x10u;
            for (int k = 0; k < SIFT3D_DESC_LEN; k++) f[i].desc[k] = (float)((k * 7 + i) % 64);
        }
        float m[4][4] = {{0.9f, 0.1f, 0.0f, -10.0f}, {-0.1f, 1.1f, 0.05f, 7.0f}, {0.0f, -0.05f, 1.4f, 3.0f}, {0, 0, 0, 1}};
        sift3d_world_transform(f, N, m);
        char p[1024];
        const char *cm[3] = {"c1", "c2", "c3"};
        snprintf(p, sizeof p, "%s/a.key", argv[2]);
        if (sift3d_write_key(p, f, N, 140.0f, 3, cm) != 0) return 2;
        sift3d_feature *g = 0;
        int64_t n = 0;
        if (sift3d_read_key(p, &g, &n) != 0 || n != N) return 3;
        for (int i = 0; i < N; i++)
            if (fabsf(g[i].x - f[i].x) > 1e-3f * (1.0f + fabsf(f[i].x)) || g[i].info != f[i].info || g[i].desc[5] != f[i].desc[5]) return 4;
        free(g);
        snprintf(p, sizeof p, "%s/a.bin", argv[2]);
        if (sift3d_write_key_bin(p, f, N, 140.0f) != 0) return 5;
        snprintf(p, sizeof p, "%s/missing/a.key", argv[2]);
        if (sift3d_write_key(p, f, N, 140.0f, 3, cm) == 0) return 6; /* a path that cannot be created fails cleanly */
        if (sift3d_read_key(p, &g, &n) == 0) return 7;
        snprintf(p, sizeof p, "%s/slice.pgm", argv[2]);
        if (sift3d_write_pgm(p, v, 16, 16) != 0) return 8;
        float flat[12];
        for (int i = 0; i < 12; i++) flat[i] = 2.5f;
        if (sift3d_write_pgm(p, flat, 3, 4) != 0) return 9; /* a constant slice: 0 / 0 in the scaling, written as zeros */
        if (sift3d_write_pgm(p, flat, 0, 4) == 0) return 10;
        printf("keys ok %d\n", N);
        free(v);
        free(f);
        return 0;
    }
    if (argc == 2 && strcmp(argv[1], "votes") == 0) {
        enum { NI = 5, PER = 40, K = 4, NF = NI * PER };
        sift3d_feature *f = (sift3d_feature *)calloc(NF, sizeof(*f));
        int8_t *bytes = (int8_t *)malloc((size_t)NF * SIFT3D_DESC_LEN);
        int64_t first[NI + 1];
        int32_t labels[NI], *idx = (int32_t *)malloc(sizeof(int32_t) * NF * K), *d2 = (int32_t *)malloc(sizeof(int32_t) * NF * K);
        float *votes = (float *)malloc(sizeof(float) * NI * 3);
        int32_t *counts = (int32_t *)malloc(sizeof(int32_t) * NI * 3);
        unsigned s = 12345u;
        for (int i = 0; i <= NI; i++) first[i] = (int64_t)i * PER;
        for (int i = 0; i < NI; i++) labels[i] = i % 3;
        for (int i = 0; i < NF; i++) {
            f[i].info = (i % 3 == 0) ? 0x30u : ((i % 3 == 1) ? 0x20u : 0x00u);
            for (int k = 0; k < SIFT3D_DESC_LEN; k++) f[i].desc[k] = (float)((k * 5 + i) % 64);
            for (int k = 0; k < K; k++) {
                s = s * 1664525u + 1013904223u;
                idx[i * K + k] = (int32_t)((s >> 8) % NF);
                d2[i * K + k] = (int32_t)(100 + 37 * k + (s & 31u));
            }
        }
        if (sift3d_match_descriptors(f, NF, bytes) != 0) return 2;
        f[7].desc[3] = 200.5f;
        if (sift3d_match_descriptors(f, NF, bytes) == 0) return 3; /* not a byte: refused */
        f[7].desc[3] = 1e30f;
        if (sift3d_match_descriptors(f, NF, bytes) == 0) return 3;
        f[7].desc[3] = 3.0f;
        if (sift3d_match_votes(f, first, NI, labels, 3, idx, d2, K, votes, counts) != 0) return 4;
        idx[17] = NF; /* one past the last feature */
        if (sift3d_match_votes(f, first, NI, labels, 3, idx, d2, K, votes, counts) == 0) return 5;
        idx[17] = -1; /* "no further neighbour": fine */
        if (sift3d_match_votes(f, first, NI, labels, 3, idx, d2, K, votes, counts) != 0) return 6;
        labels[2] = 3; /* a label beyond n_labels */
        if (sift3d_match_votes(f, first, NI, labels, 3, idx, d2, K, votes, counts) == 0) return 7;
        labels[2] = -2;
        if (sift3d_match_votes(f, first, NI, labels, 3, idx, d2, K, votes, counts) == 0) return 7;
        labels[2] = 2;
        first[2] = first[3] + 1; /* offsets that do not ascend */
        if (sift3d_match_votes(f, first, NI, labels, 3, idx, d2, K, votes, counts) == 0) return 8;
        const int64_t kept = sift3d_match_filter(f, NF, 1, 4);
        if (kept <= 0 || kept >= NF) return 9;
        printf("votes ok %lld\n", (long long)kept);
        free(f); free(bytes); free(idx); free(d2); free(votes); free(counts);
        return 0;
    }
    fprintf(stderr, "usage: host_asan_driver read <file>... | keys <dir> | votes\n");
    return 1;
}
