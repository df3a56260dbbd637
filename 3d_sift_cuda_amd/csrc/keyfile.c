/* keyfile.c -- see keyfile.h */
#include "keyfile.h"

#include <stdio.h>

static int keep(const sift3d_feature *r, float eig_thres)
{
    float es = r->eigs[0] + r->eigs[1] + r->eigs[2];
    float ep = r->eigs[0] * r->eigs[1] * r->eigs[2];
    float esp = es * es * es;
    return (esp < eig_thres * ep || eig_thres < 0);
}

int sift3d_write_key(const char *path, const sift3d_feature *recs, int64_t n, float eig_thres, int n_comments,
                     const char *const *comments)
{
    FILE *f = fopen(path, "wt");
    if (!f) return -1;
    int count = 0;
    for (int64_t i = 0; i < n; i++)
        if (keep(&recs[i], eig_thres)) count++;
    fprintf(f, "# featExtract %s\n", "1.1");
    for (int i = 0; i < n_comments; i++) fprintf(f, "# %s\n", comments[i]);
    fprintf(f, "Features: %d\n", count);
    fprintf(f, "Scale-space location[x y z scale] orientation[o11 o12 o13 o21 o22 o23 o31 o32 o32] 2nd moment "
               "eigenvalues[e1 e2 e3] info flag[i1] descriptor[d1 .. d64]\n");
    for (int64_t i = 0; i < n; i++) {
        const sift3d_feature *r = &recs[i];
        if (!keep(r, eig_thres)) continue;
        fprintf(f, "%f\t%f\t%f\t%f\t", r->x, r->y, r->z, r->scale);
        for (int j = 0; j < 9; j++) fprintf(f, "%f\t", r->ori[j]);
        for (int j = 0; j < 3; j++) fprintf(f, "%f\t", r->eigs[j]);
        fprintf(f, "%d\t", r->info);
        for (int j = 0; j < SIFT3D_DESC_LEN; j++) fprintf(f, "%i\t", (char)(r->desc[j]));
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}
