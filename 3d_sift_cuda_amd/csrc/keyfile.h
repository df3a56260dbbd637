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
/* How the parallel text writer puts its blocks into the file: 0 (default) positional writes, 1 a shared mapping of the
 * reserved file where the file system offers one (faster on some file systems, slower on the GPU boxes' overlayfs).  The bytes are
 * the same; tests and tools/key_writer_bench.c run both. */
void sift3d_write_key_mode(int mode);
/* The binary flavour, msFeature3DVectorOutputBin (MultiScale.h:228-303): the same two header lines as text
 * ("# featExtract 1.1", "Features: N"), then per kept record 4+9+3 floats, the info word and the 64 descriptor
 * values as unsigned char. */
int sift3d_write_key_bin(const char *path, const sift3d_feature *recs, int64_t n, float eig_thres);
/* Reader of the text format, msFeature3DVectorInputText (MultiScale.h:305-384): skips the '#' lines, needs the
 * "Features: N" and the column line, reads N records.  *recs is malloc'ed (release with free()).  Returns 0, -1 if
 * the file cannot be opened or has no valid header, -2 if a record is incomplete. */
int sift3d_read_key(const char *path, sift3d_feature **recs, int64_t *n);
/* 0 (default): a file in the writers' own layout (one record a line, plain decimal numbers) is parsed from memory by all threads,
 * anything else by the fscanf loop; 1: the fscanf loop always.  The records are the same bits either way (tests run both). */
void sift3d_read_key_mode(int mode);
/* ./image.pgm, the reference's debug picture: an x-y slice of floats (rows = y, cols = x) scaled to 0..255 as
 * output_float does (R/src_common/PpImageFloatOutput.cpp:131-180) and written as binary PGM the way
 * GenericImage::WriteToFile does (R/src_common/GenericImage.cpp:135-180).  Returns 0 or -1. */
int sift3d_write_pgm(const char *path, const float *slice, int rows, int cols);
#ifdef __cplusplus
}
#endif
#endif
