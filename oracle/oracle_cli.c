/*
 * oracle_cli.c -- the CPU restatement behind the reference's command line
 * (R/featExtract/featExtract.cpp:273-585, voxel-coordinate path).  TEST
 * INFRASTRUCTURE: the CPU baseline of bench.py and the golden-fixture tool.
 *
 *   featExtract_oracle [-2+|-2-] [-b|-br|-bn] <in.nii> <out.key>
 *   featExtract_oracle --synth X Y Z SEED <out.nii>      (write a blob-field volume)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nifti_min.h"
#include "sift3d_oracle.h"

void sift3d_synth_blobs(float *vol, int64_t X, int64_t Y, int64_t Z, uint32_t seed);

int main(int argc, char **argv)
{
    if (argc >= 7 && strcmp(argv[1], "--synth") == 0) {
        int64_t X = atoll(argv[2]), Y = atoll(argv[3]), Z = atoll(argv[4]);
        float *v = (float *)malloc(sizeof(float) * (size_t)(X * Y * Z));
        sift3d_synth_blobs(v, X, Y, Z, (uint32_t)strtoul(argv[5], 0, 10));
        int rc = nifti_min_write_f32(argv[6], v, (int)X, (int)Y, (int)Z, 1.0f, 1.0f, 1.0f);
        free(v);
        return rc;
    }
    int ia = 1, dbl = 0, mode = O3_DESC_SIFT;
    while (ia < argc && argv[ia][0] == '-') {
        if (argv[ia][1] == '2') dbl = (argv[ia][2] == '-') ? -1 : 1;
        else if (argv[ia][1] == 'b') mode = argv[ia][2] == 'r' ? O3_DESC_RRIEF : (argv[ia][2] == 'n' ? O3_DESC_NRRIEF : O3_DESC_BRIEF);
        else if (argv[ia][1] == 'd') { /* accepted and ignored: this is the CPU path */ }
        else { fprintf(stderr, "unknown option %s\n", argv[ia]); return -1; }
        ia++;
    }
    if (argc - ia < 2) { fprintf(stderr, "usage: featExtract_oracle [options] <in> <out>\n"); return -1; }
    nifti_min_image img;
    if (nifti_min_read(argv[ia], &img) < 0) { printf("Error: could not read input file: %s\n", argv[ia]); return -1; }
    int64_t X = img.nx, Y = img.ny, Z = img.nz;
    float *vol = img.data;
    float init_scale = 1.0f, size_factor = 1;
    if (dbl == 1) {
        float *d = (float *)malloc(sizeof(float) * (size_t)(8 * X * Y * Z));
        o3_double_size(vol, X, Y, Z, d);
        free(vol); vol = d; X *= 2; Y *= 2; Z *= 2;
        init_scale *= 0.5; size_factor /= 2;
    } else if (dbl == -1) {
        float *d = (float *)malloc(sizeof(float) * (size_t)((X / 2) * (Y / 2) * (Z / 2)));
        o3_halve_center(vol, X, Y, Z, d);
        free(vol); vol = d; X /= 2; Y /= 2; Z /= 2;
        size_factor *= 2;
    }
    if (Z <= 1) { printf("Could not read volume: %s\n", argv[ia]); return -1; }
    printf("Input image: i=%d j=%d k=%d\n", (int)X, (int)Y, (int)Z);
    o3_record *recs = 0; int64_t n = 0; o3_stats st;
    o3_extract(vol, X, Y, Z, init_scale, mode, 140.0f, size_factor, &recs, &n, &st);
    char c1[200], c2[200], c3[400];
    sprintf(c1, "Extraction Voxel Resolution (ijk) : %d %d %d", (int)X, (int)Y, (int)Z);
    sprintf(c2, "Extraction Voxel Size (mm)  (ijk) : %f %f %f", 1.0f * img.dx, 1.0f * img.dy, 1.0f * img.dz);
    sprintf(c3, "Feature Coordinate Space: voxels: 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0");
    const char *cm[3] = {c1, c2, c3};
    o3_write_key(argv[ia + 1], recs, n, 140.0f, 3, cm);
    fprintf(stderr, "records=%lld extrema=%lld keypoints=%lld octaves=%lld  blur=%.3fs dog=%.3fs sub=%.3fs detect=%.3fs feat=%.3fs desc=%.3fs\n",
            (long long)n, (long long)st.n_extrema, (long long)st.n_keypoints, (long long)st.n_octaves, st.t_blur, st.t_dog,
            st.t_subsample, st.t_detect, st.t_features, st.t_desc);
    printf("\nDone.\n");
    free(recs); free(vol);
    return 0;
}
