/*
 * match_votes.c -- feature filters, descriptor bytes and the soft-vote accumulation of the matcher (see match.h).
 * R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/.
 */
#include "match.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int64_t sift3d_match_filter(sift3d_feature *f, int64_t n, int reoriented, int peaks)
{
    int64_t kept = 0;
    for (int64_t i = 0; i < n; i++) {
        const int is_re = (f[i].info & SIFT3D_INFO_REORIENT) != 0, is_valley = (f[i].info & SIFT3D_INFO_MIN0MAX1) != 0;
        if (reoriented ? !is_re : is_re) continue;
        if (peaks == 0 && is_valley) continue;  /* removeNonPeakFeatures drops the records with the flag */
        if (peaks == 1 && !is_valley) continue; /* removeNonValleyFeatures drops those without */
        sift3d_feature r = f[i];
        if (!reoriented) { /* removeReorientedFeatures resets the frame of what it keeps */
            memset(r.ori, 0, sizeof r.ori);
            r.ori[0] = r.ori[4] = r.ori[8] = 1.0f;
        }
        f[kept++] = r;
    }
    return kept;
}

int sift3d_match_descriptors(const sift3d_feature *f, int64_t n, int8_t *out)
{
    /* The reference hands the floats to FLANN as they are; the rank transform leaves whole numbers 0..63 there.  The search
     * here works on bytes, so anything that is not a whole number in 0..127 is refused -- tested on the float, because a
     * cast of an out-of-range float to char is undefined (advisor, round 3) and would wrap some of them into range. */
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < SIFT3D_DESC_LEN; j++) {
            const float d = f[i].desc[j];
            if (!(d >= 0.0f && d <= 127.0f) || d != (float)(int)d) return -1;
            out[i * SIFT3D_DESC_LEN + j] = (int8_t)(int)d;
        }
    return 0;
}

/* which database feature a query image's features have voted for already, and with what weight: open addressing over
 * the feature index (the reference keeps a std::map per query image, featMatchUtilities.cpp:1620) */
typedef struct {
    int32_t *key;
    float *val;
    int64_t cap, used;
} vote_map;

static int vm_init(vote_map *m, int64_t expect)
{
    m->cap = 64;
    while (m->cap < 2 * expect + 16) m->cap *= 2;
    m->key = (int32_t *)malloc(sizeof(int32_t) * (size_t)m->cap);
    m->val = (float *)malloc(sizeof(float) * (size_t)m->cap);
    m->used = 0;
    if (!m->key || !m->val) return -1;
    memset(m->key, 0xff, sizeof(int32_t) * (size_t)m->cap);
    return 0;
}
static float *vm_find(vote_map *m, int32_t k, int create)
{
    int64_t h = (int64_t)(((uint64_t)(uint32_t)k * 0x9E3779B97F4A7C15ull) >> 20) & (m->cap - 1);
    while (m->key[h] != -1) {
        if (m->key[h] == k) return &m->val[h];
        h = (h + 1) & (m->cap - 1);
    }
    if (!create) return NULL;
    m->key[h] = k;
    m->used++;
    return &m->val[h];
}

/* the votes of one query image (row img of votes / counts); scratch: 2k int32 + 2k float.  Returns 0, -1 out of memory */
static int votes_of_image(int img, const int64_t *first, const int32_t *img_of, const int32_t *labels, int n_labels,
                          const int32_t *nn_idx, const int32_t *nn_dist2, int k, float *votes, int32_t *counts, int32_t *iscratch,
                          float *fscratch)
{
    int32_t *acc_idx = iscratch, *seen_img = iscratch + k;
    float *acc_dist = fscratch, *w = fscratch + k;
    const int64_t lo = first[img], cnt = first[img + 1] - first[img];
    vote_map vm;
    if (vm_init(&vm, cnt * k)) {
        free(vm.key);
        free(vm.val);
        return -1;
    }
    for (int64_t q = lo; q < lo + cnt; q++) {
        /* the neighbours from other images, one per image, nearest first; the smallest positive distance among them */
        int na = 0;
        float min_dist = -1.0f;
        for (int j = 0; j < k; j++) {
            const int32_t r = nn_idx[q * k + j];
            if (r < 0) break;
            /* "must not be from the query image": the reference's test is r < lo || r > lo + cnt (:1660), which also
             * drops index lo + cnt, the first feature of the next image */
            if (!(r < lo || r > lo + cnt)) continue;
            int dup = 0;
            for (int a = 0; a < na; a++) dup |= seen_img[a] == img_of[r];
            if (dup) continue;
            const float d = (float)nn_dist2[q * k + j]; /* FLANN hands back squared distances; the reference calls them distances */
            acc_idx[na] = r;
            acc_dist[na] = d;
            if ((min_dist == -1.0f || d < min_dist) && d > 0) min_dist = d;
            seen_img[na++] = img_of[r];
        }
        float sum = 0.0f;
        for (int a = 0; a < na; a++) { /* appearance weight, :1702-1716 */
            const float dsq = acc_dist[a] * acc_dist[a], var = min_dist * min_dist;
            w[a] = expf(-dsq / var);
            sum += w[a];
        }
        if (sum <= 0) continue;
        for (int a = 0; a < na; a++) { /* soft max + log with background 1, :1726-1735 */
            w[a] /= sum;
            w[a] += 1.0f;
            w[a] = logf(w[a]);
            w[a] /= logf(2.0f);
        }
        for (int a = 0; a < na; a++) { /* one vote per database feature and query image: the better one stays, :1779-1802 */
            const int label = labels[img_of[acc_idx[a]]];
            float *slot = &votes[(size_t)img * (size_t)n_labels + (size_t)label];
            float *prev = vm_find(&vm, acc_idx[a], 0);
            if (prev) {
                if (w[a] > *prev) {
                    if (*prev > 0) *slot -= *prev;
                    *slot += w[a];
                    *prev = w[a];
                }
            } else {
                *slot += w[a];
                counts[(size_t)img * (size_t)n_labels + (size_t)label] += 1;
                *vm_find(&vm, acc_idx[a], 1) = w[a];
            }
        }
    }
    free(vm.key);
    free(vm.val);
    return 0;
}

int sift3d_match_votes(const sift3d_feature *feats, const int64_t *first, int n_images, const int32_t *labels, int n_labels,
                       const int32_t *nn_idx, const int32_t *nn_dist2, int k, float *votes, int32_t *counts)
{
    (void)feats; /* positions enter the reference's loop only through a location gate that is switched off (:1619) */
    if (!first || !labels || !nn_idx || !nn_dist2 || !votes || !counts || n_images < 1 || k < 1 || n_labels < 1) return -1;
    const int64_t total = first[n_images];
    if (first[0] != 0 || total < 0 || total > INT32_MAX) return -1;
    for (int i = 0; i < n_images; i++)
        if (first[i + 1] < first[i] || labels[i] < 0 || labels[i] >= n_labels) return -1; /* votes[] / counts[] are indexed by label */
    for (int64_t e = 0; e < total * k; e++)
        if (nn_idx[e] >= total) return -1; /* img_of[] is indexed by neighbour; negative = "no further neighbour" */
    int rc = 0;
    int32_t *img_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    if (!img_of) return -1;
    for (int i = 0; i < n_images; i++)
        for (int64_t f = first[i]; f < first[i + 1]; f++) img_of[f] = i;
    memset(votes, 0, sizeof(float) * (size_t)n_images * (size_t)n_labels);
    memset(counts, 0, sizeof(int32_t) * (size_t)n_images * (size_t)n_labels);
    /* Query images are independent -- each writes its own row of votes / counts and keeps its own map -- and the reference
     * runs this loop over images with OpenMP (featMatchMultiple.cpp:108).  Within an image the order is the serial one, so
     * the rows are the same bytes for any thread count. */
    int vote_threads = 1;
#ifdef _OPENMP
    vote_threads = omp_get_max_threads(); /* (capped: a box of the pool shows 256 cores to a 16-CPU share) */
    if (vote_threads > 32) vote_threads = 32;
    if (vote_threads > n_images) vote_threads = n_images > 0 ? n_images : 1;
#pragma omp parallel num_threads(vote_threads)
#endif
    {
        int32_t *iscratch = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)k);
        float *fscratch = (float *)malloc(sizeof(float) * 2 * (size_t)k);
        int bad = !iscratch || !fscratch;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int img = 0; img < n_images; img++)
            if (!bad && votes_of_image(img, first, img_of, labels, n_labels, nn_idx, nn_dist2, k, votes, counts, iscratch, fscratch)) bad = 1;
        if (bad) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            rc = -1;
        }
        free(iscratch);
        free(fscratch);
    }
    free(img_of);
    return rc;
}

int sift3d_match_write_votes(const char *votes_path, const char *counts_path, const char *title, const float *votes,
                             const int32_t *counts, int n_images, int n_labels, int append)
{
    FILE *fv = fopen(votes_path, append ? "at" : "wt"), *fc = fopen(counts_path, append ? "at" : "wt");
    if (!fv || !fc) {
        if (fv) fclose(fv);
        if (fc) fclose(fc);
        return -1;
    }
    fprintf(fv, "%s\n", title);
    fprintf(fc, "%s\n", title);
    for (int i = 0; i < n_images; i++) {
        for (int j = 0; j < n_labels; j++) {
            fprintf(fv, "%f\t", votes[(size_t)i * (size_t)n_labels + (size_t)j]);
            fprintf(fc, "%d\t", counts[(size_t)i * (size_t)n_labels + (size_t)j]);
        }
        fprintf(fv, "\n");
        fprintf(fc, "\n");
    }
    fprintf(fv, "\n");
    fprintf(fc, "\n");
    fclose(fv);
    fclose(fc);
    return 0;
}
