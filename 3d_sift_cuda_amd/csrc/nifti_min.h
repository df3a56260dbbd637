/*
 * nifti_min.h -- minimal NIfTI-1 / Analyze-7.5 volume I/O for featExtract.
 *
 * Replaces, for this path only, what the reference gets from the vendored
 * niftilib through fioReadNifti (R/featExtract/featExtract.cpp:84-220, where
 * R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/): read the
 * header, load the voxels (.nii, .nii.gz, .hdr + .img[.gz]), cast them to
 * float32 with a plain C cast (scl_slope / scl_inter ignored, as
 * reg_changeDatatype does, featExtract.cpp:18-77), and expose the qform /
 * sform matrices needed by the -w / -ws options.
 */
#ifndef NIFTI_MIN_H
#define NIFTI_MIN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int nx, ny, nz, nt;
    float dx, dy, dz;
    int datatype;      /* NIfTI datatype code of the file */
    int qform_code, sform_code;
    float qto_xyz[4][4];
    float sto_xyz[4][4];
    float *data;       /* nx*ny*nz*nt float32, x fastest; free() it */
} nifti_min_image;

/* Returns 0 on success; -1 cannot open/parse header, -2 no voxel data,
 * -3 unsupported datatype, -4 out of memory. */
int nifti_min_read(const char *path, nifti_min_image *img);
/* The same in two steps, for a caller that wants the voxels as they arrive (featExtract uploads the planes it has while the
 * rest of a .nii.gz is still being inflated): nifti_min_open parses the header into *img (data stays NULL) and leaves the
 * stream at the first voxel; nifti_min_read_voxels casts the next nvox voxels into dst (0, -2 short file, -4 memory);
 * nifti_min_close releases the stream.  nifti_min_read is open + read everything + close. */
typedef struct nifti_min_stream nifti_min_stream;
int nifti_min_open(const char *path, nifti_min_image *img, nifti_min_stream **s);
int nifti_min_read_voxels(nifti_min_stream *s, float *dst, size_t nvox);
void nifti_min_close(nifti_min_stream *s);
/* 0: gzip'ed files through zlib's streaming inflate only; 1 (default): in one call through libdeflate where the system has it and the
 * file is one gzip member of at most 4 GiB (nifti_min.c: fast_inflate).  The voxels are the same. */
void nifti_min_fast_inflate(int on);
int nifti_min_fast_inflate_count(void); /* files that went through libdeflate so far in this process */
/* Writes a float32 single-file .nii (or .nii.gz by extension), voxel size (dx,dy,dz). */
int nifti_min_write_f32(const char *path, const float *data, int nx, int ny, int nz, float dx, float dy, float dz);
/* As above with a qform (quaternion b,c,d, offsets, qfac = pixdim[0]) and/or an sform (3 rows of 4):
 * pass NULL to leave the corresponding code 0.  For tests of the -w / -ws options. */
int nifti_min_write_f32_ex(const char *path, const float *data, int nx, int ny, int nz, float dx, float dy, float dz,
                           const float *quatern_bcd_xyz_qfac /* 7 floats */, const float *srow /* 12 floats */);
void nifti_min_free(nifti_min_image *img);

#ifdef __cplusplus
}
#endif
#endif
