/*
 * match_oracle.c -- CPU restatement of the matcher's arithmetic (TEST INFRASTRUCTURE: only tests/, bench tools and
 * __graft_entry__.smoke() may use it; the product never links it).
 *
 * What is restated, R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/:
 *   o3_knn64        the search flann_find_nearest_neighbors_index answers approximately
 *                   (R/feat_common/featMatchUtilities.cpp:1612), done by brute force with the distance of
 *                   Feature3DInfo::DistSqrPCs (R/src_common/MultiScale.h:61-73: sum of squared component differences);
 *   o3_match_votes  msNearestNeighborApproximateSearchSelf's vote accumulation (:1584-1819), the part that ends in
 *                   matching_votes.txt / vote_count.txt (matchAllToAll, R/featMatchMultiple/featMatchMultiple.cpp:118-139).
 * PARITY UNPINNED against the reference: FLANN (the kd-tree search) is an un-vendored dependency fetched by the
 * reference's CMake (R/CMakeLists.txt:67-187, not under /root/reference), its search is approximate and randomised, and
 * the reference holds no fixture for this path.  What these functions pin is the product against an independent
 * statement of the same published arithmetic: exact nearest neighbours under (distance, index) order, and the votes
 * that follow from them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* k nearest database vectors of every query, ascending by (squared distance, index); -1 / INT32_MAX past the end */
int o3_knn64(const int8_t *db, int64_t n_db, const int8_t *q, int64_t n_q, int k, int32_t *idx, int32_t *dist2)
{
    if (!db || !q || !idx || !dist2 || k < 1) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < n_q; i++) {
        int32_t *bi = idx + i * k, *bd = dist2 + i * k;
        for (int j = 0; j < k; j++) {
            bi[j] = -1;
            bd[j] = INT32_MAX;
        }
        for (int64_t b = 0; b < n_db; b++) {
            int32_t d = 0;
            for (int c = 0; c < 64; c++) { /* DistSqrPCs */
                const int32_t diff = (int32_t)q[i * 64 + c] - (int32_t)db[b * 64 + c];
                d += diff * diff;
            }
            if (d >= bd[k - 1]) continue; /* indices ascend: a tie with the k-th stays out */
            int j = k - 1;
            while (j > 0 && bd[j - 1] > d) {
                bd[j] = bd[j - 1];
                bi[j] = bi[j - 1];
                j--;
            }
            bd[j] = d;
            bi[j] = (int32_t)b;
        }
    }
    return 0;
}

/* votes[img][label], counts[img][label]: ppfMatchingVotes / ppiLabelVotes of matchAllToAll */
int o3_match_votes(const int64_t *first, int n_images, const int32_t *labels, int n_labels, const int32_t *nn_idx,
                   const int32_t *nn_dist2, int k, float *votes, int32_t *counts)
{
    const int64_t total = first[n_images];
    int32_t *img_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total + 1));
    float *voted = (float *)malloc(sizeof(float) * (size_t)(total + 1)); /* votedFeatures of the query image (:1620) as a dense array */
    char *has = (char *)malloc((size_t)(total + 1));
    if (!img_of || !voted || !has) return -1;
    for (int i = 0; i < n_images; i++)
        for (int64_t f = first[i]; f < first[i + 1]; f++) img_of[f] = i;
    for (int64_t i = 0; i < (int64_t)n_images * n_labels; i++) {
        votes[i] = 0.0f;
        counts[i] = 0;
    }
    for (int img = 0; img < n_images; img++) {
        const int64_t start = first[img], cnt = first[img + 1] - first[img]; /* g_piImgIndices, g_piImgFeatureCounts */
        memset(has, 0, (size_t)(total + 1));
        for (int64_t i = 0; i < cnt; i++) {
            const int64_t qf = start + i;
            int32_t nn_i[64];
            float nn_d[64], weights[64];
            int images[64], n = 0;
            float fMinDist = -1; /* :1634 */
            for (int j = 0; j < k && j < 64; j++) {
                const int32_t r = nn_idx[qf * k + j];
                if (r < 0) break;
                if (r < start || r > start + cnt) { /* :1660 */
                    int found = 0;
                    for (int m = 0; m < n; m++) found |= images[m] == img_of[r]; /* :1666 */
                    if (!found) {
                        const float dist = (float)nn_dist2[qf * k + j];
                        nn_i[n] = r;
                        nn_d[n] = dist;
                        if (fMinDist == -1 || dist < fMinDist) { /* :1672-1679 */
                            if (dist > 0) fMinDist = dist;
                        }
                        images[n++] = img_of[r];
                    }
                }
            }
            float fSumWeights = 0;
            for (int j = 0; j < n; j++) { /* :1693-1719 */
                const float fDistApp = nn_d[j], fDistSqApp = fDistApp * fDistApp, fVarApp = fMinDist * fMinDist;
                const float fAppWeight = expf(-fDistSqApp / fVarApp);
                weights[j] = fAppWeight;
                fSumWeights += fAppWeight;
            }
            if (fSumWeights <= 0) continue; /* :1722 */
            const float eta = 1;
            for (int j = 0; j < n; j++) { /* :1727-1735 */
                weights[j] /= fSumWeights;
                weights[j] += eta;
                weights[j] = logf(weights[j]);
                weights[j] /= logf(eta + 1);
            }
            for (int j = 0; j < n; j++) { /* :1738-1805 */
                const int32_t r = nn_i[j];
                const int label = labels[img_of[r]];
                float *cell = &votes[(size_t)img * (size_t)n_labels + (size_t)label];
                if (has[r]) {
                    const float previous = voted[r];
                    if (weights[j] > previous) {
                        if (previous > 0) *cell -= previous;
                        *cell += weights[j];
                        voted[r] = weights[j];
                    }
                } else {
                    *cell += weights[j];
                    counts[(size_t)img * (size_t)n_labels + (size_t)label] += 1;
                    has[r] = 1;
                    voted[r] = weights[j];
                }
            }
        }
    }
    free(img_of); free(voted); free(has);
    return 0;
}
