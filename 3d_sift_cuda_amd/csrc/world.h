/*
 * world.h -- the -w / -ws options of featExtract: isotropic resampling of the input and the transform of
 * the records to world (millimetre) coordinates.  Host-side, as in the reference
 * (R/featExtract/featExtract.cpp:118-204, 429-538; R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/).
 */
#ifndef SIFT3D_WORLD_H
#define SIFT3D_WORLD_H
#include "nifti_min.h"
#include "sift3d.h"
#ifdef __cplusplus
extern "C" {
#endif
/* If the voxel sizes differ: resample img->data to isotropic voxels of the smallest size (trilinear,
 * featExtract.cpp:124-198), rescale the columns of qto_xyz / sto_xyz and set dx = dy = dz.  Returns 0,
 * or -3 when the resampled volume cannot be allocated. */
int sift3d_world_make_isotropic(nifti_min_image *img);
/* Records (voxel units, size factor already applied) -> world coordinates with the 4x4 matrix m
 * (featExtract.cpp:447-538). */
void sift3d_world_transform(sift3d_feature *recs, int64_t n, float m[4][4]);
#ifdef __cplusplus
}
#endif
#endif
