/*
 * oracle_cli.c -- the CPU restatement behind the reference's command line
 * (R/featExtract/featExtract.cpp:94-214 and 273-585).  TEST
 * INFRASTRUCTURE: the CPU baseline of bench.py and the golden-fixture tool.
 *
 *   featExtract_oracle [-2+|-2-] [-w|-ws] [-b|-br|-bn] <in.nii> <out.key>
 *   featExtract_oracle --synth X Y Z SEED <out.nii> [dx dy dz [b c d qx qy qz qfac [srow x12]]]
 *                                      (write a blob-field volume, optionally with voxel sizes, a qform, an sform)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nifti_min.h"
#include "sift3d_oracle.h"

void sift3d_synth_blobs(float *vol, int64_t X, int64_t Y, int64_t Z, uint32_t seed);

/* fioReadNifti's isotropic branch, featExtract.cpp:118-198: resample to the smallest voxel size */
static int resample_isotropic(nifti_min_image *im)
{
    if (im->dx == im->dy && im->dy == im->dz) return 0;
    float s = im->dx;
    if (im->dy < s) s = im->dy;
    if (im->dz < s) s = im->dz;
    int X = (int)(im->nx * im->dx / s), Y = (int)(im->ny * im->dy / s), Z = (int)(im->nz * im->dz / s);
    float k[3] = {s / im->dx, s / im->dy, s / im->dz};
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { /* featExtract.cpp:162-171 */
            im->qto_xyz[r][c] *= k[c];
            if (im->sform_code > 0) im->sto_xyz[r][c] *= k[c];
        }
    float *o = (float *)malloc(sizeof(float) * (size_t)X * Y * Z), *w = o;
    if (!o) return -3;
    for (int z = 0; z < Z; z++)
        for (int y = 0; y < Y; y++)
            for (int x = 0; x < X; x++) {
                double sx = x * k[0] + 0.5, sy = y * k[1] + 0.5, sz = z * k[2] + 0.5; /* float product, double sum */
                *w++ = o3_trilinear(im->data, im->nx, im->ny, im->nz, (float)sx, (float)sy, (float)sz);
            }
    free(im->data);
    im->data = o; im->nx = X; im->ny = Y; im->nz = Z;
    im->dx = im->dy = im->dz = s;
    return 0;
}

/* invert_3x3<float,float>, MultiScale.h:192-222 (the <float,double> flavour is o3_invert3) */
static void inv3f(float a[3][3], float o[3][3])
{
#define C2(r0, c0, r1, c1) (a[r0][c0] * a[r1][c1])
    float det = a[0][0] * (C2(2, 2, 1, 1) - C2(1, 2, 2, 1)) - a[1][0] * (C2(2, 2, 0, 1) - C2(0, 2, 2, 1)) +
                a[2][0] * (C2(1, 2, 0, 1) - C2(1, 1, 0, 2));
    float d = 1 / det;
    o[0][0] = (C2(2, 2, 1, 1) - C2(1, 2, 2, 1)) * d;
    o[1][0] = -(C2(2, 2, 1, 0) - C2(1, 2, 2, 0)) * d;
    o[2][0] = (C2(2, 1, 1, 0) - C2(1, 1, 2, 0)) * d;
    o[0][1] = -(C2(2, 2, 0, 1) - C2(0, 2, 2, 1)) * d;
    o[1][1] = (C2(2, 2, 0, 0) - C2(0, 2, 2, 0)) * d;
    o[2][1] = -(C2(2, 1, 0, 0) - C2(0, 1, 2, 0)) * d;
    o[0][2] = (C2(1, 2, 0, 1) - C2(0, 2, 1, 1)) * d;
    o[1][2] = -(C2(1, 2, 0, 0) - C2(0, 2, 1, 0)) * d;
    o[2][2] = (C2(1, 1, 0, 0) - C2(0, 1, 1, 0)) * d;
#undef C2
}

/* featExtract.cpp:436-470 and 507-538 */
static void to_world(o3_record *r, int64_t n, float M[4][4])
{
    float R[3][3], sc = 0;
    for (int i = 0; i < 3; i++) {
        float q = M[i][0] * M[i][0] + M[i][1] * M[i][1] + M[i][2] * M[i][2];
        float mag = q > 0 ? sqrtf(q) : 0; /* vec3D_mag */
        sc += mag;
        if (q > 0) { /* vec3D_norm_3d: float fDiv = 1.0 / sqrt(float) */
            float dv = (float)(1.0 / sqrtf(q));
            for (int j = 0; j < 3; j++) R[i][j] = M[i][j] * dv;
        } else {
            R[i][0] = 1; R[i][1] = 0; R[i][2] = 0;
        }
    }
    sc /= 3;
    for (int64_t t = 0; t < n; t++, r++) {
        float p[4] = {r->x, r->y, r->z, 1}, q[3];
        for (int i = 0; i < 3; i++) {
            float acc = 0;
            for (int j = 0; j < 4; j++) acc += M[i][j] * p[j];
            q[i] = acc;
        }
        r->x = q[0]; r->y = q[1]; r->z = q[2];
        r->scale *= sc;
        float o[3][3], oi[3][3], ro[3][3];
        memcpy(o, r->ori, sizeof o);
        inv3f(o, oi);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                float acc = 0;
                for (int m = 0; m < 3; m++) acc += R[i][m] * oi[m][j];
                ro[i][j] = acc;
            }
        inv3f(ro, o);
        memcpy(r->ori, o, sizeof o);
    }
}

int main(int argc, char **argv)
{
    if (argc >= 7 && strcmp(argv[1], "--synth") == 0) {
        int64_t X = atoll(argv[2]), Y = atoll(argv[3]), Z = atoll(argv[4]);
        float *v = (float *)malloc(sizeof(float) * (size_t)(X * Y * Z));
        sift3d_synth_blobs(v, X, Y, Z, (uint32_t)strtoul(argv[5], 0, 10));
        float a[22] = {1, 1, 1};
        int na = argc - 7 > 22 ? 22 : argc - 7;
        for (int i = 0; i < na; i++) a[i] = (float)atof(argv[7 + i]);
        int rc = nifti_min_write_f32_ex(argv[6], v, (int)X, (int)Y, (int)Z, a[0], a[1], a[2], na >= 10 ? a + 3 : 0, na >= 22 ? a + 10 : 0);
        free(v);
        return rc;
    }
    int ia = 1, dbl = 0, mode = O3_DESC_SIFT, world = 0;
    while (ia < argc && argv[ia][0] == '-') {
        if (argv[ia][1] == '2') dbl = (argv[ia][2] == '-') ? -1 : 1;
        else if (argv[ia][1] == 'b') mode = argv[ia][2] == 'r' ? O3_DESC_RRIEF : (argv[ia][2] == 'n' ? O3_DESC_NRRIEF : O3_DESC_BRIEF);
        else if (argv[ia][1] == 'w') world = argv[ia][2] == 's' ? 2 : 1;
        else if (argv[ia][1] == 'd') { /* accepted and ignored: this is the CPU path */ }
        else { fprintf(stderr, "unknown option %s\n", argv[ia]); return -1; }
        ia++;
    }
    if (argc - ia < 2) { fprintf(stderr, "usage: featExtract_oracle [options] <in> <out>\n"); return -1; }
    nifti_min_image img;
    if (nifti_min_read(argv[ia], &img) < 0) { printf("Error: could not read input file: %s\n", argv[ia]); return -1; }
    if (world && resample_isotropic(&img) < 0) { printf("Error: could not read input file: %s\n", argv[ia]); return -1; }
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
    if (world) {
        float(*M)[4] = img.qto_xyz;
        if (world == 2) {
            if (img.sform_code > 0) M = img.sto_xyz;
            else printf("Error: sform_code <= 0, output to qto_xyz instead of sto_xyz");
        }
        to_world(recs, n, M);
        int k = sprintf(c3, "Feature Coordinate Space: millimeters (%s) :", world == 2 ? "sto_xyz" : "qto_xyz");
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 4; j++) k += sprintf(c3 + k, " %f", M[i][j]);
        sprintf(c3 + k, " 0.0 0.0 0.0 1.0");
    } else
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
