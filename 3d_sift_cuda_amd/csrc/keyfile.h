/*
 * keyfile.h -- the ".key" text output of featExtract.
 * Replaces msFeature3DVectorOutputText (R/src_common/MultiScale.h:386-474, R/ =
 * /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/): same header
 * lines, same column line, same printf formats, same eigenvalue filter.
 */
#ifndef SIFT3D_KEYFILE_H
#define SIFT3D_KEYFILE_H
#include "sift3d.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Returns 0, or -1 if the file cannot be opened.  eig_thres < 0 keeps every record. */
int sift3d_write_key(const char *path, const sift3d_feature *recs, int64_t n, float eig_thres, int n_comments,
                     const char *const *comments);
#ifdef __cplusplus
}
#endif
#endif
