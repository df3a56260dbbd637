/*
 * match.h -- host side of the matcher: what featMatchMultiple's all-to-all mode does with the
 * nearest-neighbour lists (R/featMatchMultiple/featMatchMultiple.cpp:18-146 matchAllToAll,
 * R/feat_common/featMatchUtilities.cpp:1584-1819 msNearestNeighborApproximateSearchSelf; R/ =
 * /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/).  The lists come from sift3d_knn64 (exact, on the GPU);
 * the vote accumulation is host arithmetic in the reference and stays host arithmetic here.
 */
#ifndef SIFT3D_MATCH_H
#define SIFT3D_MATCH_H
#include "sift3d.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The feature filters of the reference's main (featMatchMultiple.cpp:602-623; featMatchUtilities.cpp:1263-1340), applied in
 * place; returns the number of features kept.  reoriented: 1 keep only records with INFO_FLAG_REORIENT, 0 keep only those
 * without (and reset their frames to the identity); peaks: 0 peaks only (flag MIN0MAX1 clear), 1 valleys only (flag set),
 * anything else both. */
int64_t sift3d_match_filter(sift3d_feature *f, int64_t n, int reoriented, int peaks);
/* Descriptor of a record as the 64 signed bytes sift3d_knn64 takes: the (char) of each value, which is what a .key file
 * holds (MultiScale.h:447-455).  Returns -1 if a value is outside 0..127. */
int sift3d_match_descriptors(const sift3d_feature *f, int64_t n, int8_t *out);
/* Soft votes between images.  All features of all images back to back (image i = features first[i] .. first[i+1]-1,
 * n_images + 1 entries), labels[i] the label of image i (the reference uses the image index), n_labels > every label.
 * nn_idx / nn_dist2: the k nearest neighbours of every feature among ALL features, ascending (sift3d_knn64 with the same
 * array as database and queries).  votes / counts: n_images x n_labels, row = image of the query feature, column = label
 * of the matched feature: ppfMatchingVotes / ppiLabelVotes.  Returns 0. */
int sift3d_match_votes(const sift3d_feature *feats, const int64_t *first, int n_images, const int32_t *labels, int n_labels,
                       const int32_t *nn_idx, const int32_t *nn_dist2, int k, float *votes, int32_t *counts);
/* matching_votes.txt / vote_count.txt as matchAllToAll writes them (featMatchMultiple.cpp:118-139): the title line, one
 * row per image with tab-separated %f / %d, an empty line.  append != 0: "at" instead of "wt" (the -s2 mode). */
int sift3d_match_write_votes(const char *votes_path, const char *counts_path, const char *title, const float *votes,
                             const int32_t *counts, int n_images, int n_labels, int append);
#ifdef __cplusplus
}
#endif
#endif
