/*
 * sift3d.h -- C-ABI of the MI355X-native 3D SIFT extraction path.
 *
 * Drop-in boundary for the accelerator back-end of CarluerJB/3D_SIFT_CUDA's
 * featExtract.  Plain C types only.  Each entry point names the reference
 * interface it replaces; R/ stands for
 * /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/.
 *
 * Conventions (differences from the reference are deliberate and listed in
 * INTEGRATION.md): every function returns SIFT3D_OK (0) or a negative
 * sift3d_status instead of calling exit() like gpuErrchk
 * (R/cuda_common/SIFT_cuda_Tools.cuh:13-21); sizes are int64_t; volumes are
 * dense float32, x fastest, index (z*ny + y)*nx + x (FEATUREIO,
 * R/src_common/FeatureIO.h:21-33); the caller owns every buffer it passes
 * in; arrays the library returns are released with sift3d_free().
 * Results are those of the reference's CPU path (-d omitted): pass order
 * x,y,z, zero borders, separate multiply/add in ascending tap order.
 *
 * There is no CPU fallback: without a usable HIP device sift3d_create()
 * returns NULL and sift3d_device_count() returns 0.
 *
 * THE CALLS THAT MATTER to a maintainer of the reference (everything else in
 * this header serves tests, measurements, several GPUs, or the matcher):
 *   lifecycle     sift3d_device_count, sift3d_create, sift3d_destroy,
 *                 sift3d_last_error, sift3d_free
 *   the four accelerator wrappers of R/cuda_common/SIFT_cuda_Tools.cuh
 *                 sift3d_gauss_blur   <- blur_3d_simpleborders_CUDA_Row_Col_Shared_mem
 *                 sift3d_dog          <- fioCudaMultSum
 *                 sift3d_subsample2   <- SubSampleInterpolateCuda
 *                 sift3d_extrema      <- detectExtrema4D_test_cuda
 *   the pipeline  sift3d_set_volume (or _resized for -2+ / -2-), sift3d_extract
 *                 <- msGeneratePyramidDOG3D_efficient + the descriptor loop
 *   output        sift3d_write_key (csrc/keyfile.h, libsift3d_host.so) <- msFeature3DVectorOutputText
 * INTEGRATION.md shows the reference-side edit for each.  Section index:
 * operator level; pipeline level; tuning; timing; Z-slab building blocks and
 * sift3d_extract_zslab (several GPUs, beyond the reference); host helpers
 * (.key, NIfTI, world coordinates); matcher.  Development hooks and the one
 * hardware self-test are in sift3d_dev.h, not here.
 */
#ifndef SIFT3D_H
#define SIFT3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout version of the structures this header passes by pointer (sift3d_zslab_stats, sift3d_timings, sift3d_feature, ...).
 * The library writes WHOLE structures through the caller's pointers, so a binding compiled against another layout would be
 * overrun: a binding checks sift3d_abi_version() == SIFT3D_ABI_VERSION once, when it loads the library (the in-tree ctypes
 * mirror and featExtract do).  6: the tuning enum gained SIFT3D_TUNE_FUSED_STAGGER, sift3d_set_libm_variant (round 6); 5: sift3d_zslab_stats gained comm_sets, resident_volume, merge_ms, halo_bytes_subsample, enqueue_ms (round 5); 4: transport,
 * transport_fell_back, rccl_version (round 4). */
#define SIFT3D_ABI_VERSION 6
int sift3d_abi_version(void);

#define SIFT3D_DESC_LEN 64
#define SIFT3D_INFO_MIN0MAX1 0x00000010u /* R/src_common/MultiScale.h:28 */
#define SIFT3D_INFO_REORIENT 0x00000020u /* R/src_common/MultiScale.h:30 */

typedef enum {
    SIFT3D_OK = 0,
    SIFT3D_ERR_ARG = -1,      /* bad argument / shape */
    SIFT3D_ERR_DEVICE = -2,   /* HIP runtime error (sift3d_last_error has the text) */
    SIFT3D_ERR_MEMORY = -3,   /* host or device allocation failed */
    SIFT3D_ERR_CAPACITY = -4, /* caller-provided output array too small (counts are still exact) */
    SIFT3D_ERR_COMM = -5      /* a halo copy between slabs (sift3d_extract_zslab) failed */
} sift3d_status;

/* Descriptor selected by -b / -br / -bn (/root/reference/README.md:26-34;
 * alternatives at R/src_common/MultiScale.cpp:1037-1045). */
typedef enum { SIFT3D_DESC_SIFT = 0, SIFT3D_DESC_BRIEF = 1, SIFT3D_DESC_RRIEF = 2, SIFT3D_DESC_NRRIEF = 3 } sift3d_desc_mode;

/* LOCATION_VALUE_XYZ, R/src_common/LocationValue.h:41-47 */
typedef struct {
    int32_t x, y, z;
    float value;
} sift3d_extremum;

/* What msFeature3DVectorOutputText prints per record
 * (R/src_common/MultiScale.h:386-474): Feature3DInfo without the patch. */
typedef struct {
    float x, y, z, scale;
    float ori[9];
    float eigs[3];
    uint32_t info;
    float desc[SIFT3D_DESC_LEN];
} sift3d_feature;

/* One validated scale-space extremum before the per-keypoint stage. */
typedef struct {
    int32_t octave, level; /* level 1..3: DoG index inside the octave */
    int32_t is_max;
    int32_t x, y, z;
    float value, h_value, l_value; /* DoG at the extremum, one level below, one above */
} sift3d_candidate;

typedef struct sift3d_ctx sift3d_ctx;

/* ---- devices and contexts --------------------------------------------------
 * get_num_device()/check_best_device(), R/featExtract/featExtract.cpp:238-270;
 * the device-buffer life cycle buried in fioAllocate/fioDelete/fioCopy
 * (R/src_common/FeatureIO.cpp:384-387,527-530,1857-1860). */
int sift3d_device_count(void);
/* Allocates the device-resident pyramid for volumes of up to nx*ny*nz voxels on
 * HIP device `device` (0-based, as cudaSetDevice receives it in the reference).
 * NULL on failure. */
sift3d_ctx *sift3d_create(int device, int64_t nx, int64_t ny, int64_t nz);
/* A context for the Z-slab building blocks below only: the caller owns every level buffer, so none is allocated here
 * (a twelfth of the memory of sift3d_create).  nz_local: the most slices of an nx * ny volume any one call will be handed
 * (slab + halos).  The *_dev operators, sift3d_candidates_reset / _extrema_append*_dev / _candidates_dev / _describe_dev
 * work; entry points that use the context's own pyramid (host-pointer operators, sift3d_set_volume*, sift3d_detect,
 * sift3d_extract*) return SIFT3D_ERR_ARG. */
sift3d_ctx *sift3d_create_slab(int device, int64_t nx, int64_t ny, int64_t nz_local);
void sift3d_destroy(sift3d_ctx *ctx);
const char *sift3d_last_error(const sift3d_ctx *ctx);
/* Optional: run on a caller-owned hipStream_t (e.g. one of torch's); NULL = the
 * context's own stream.  Ordering of device buffers handed to the *_dev entry
 * points: while the context runs on its own stream, every *_dev call is fenced
 * against the legacy default stream in both directions with events (it waits for
 * what the default stream has queued, and the default stream waits for what the
 * call queued), so a caller that works on the default stream -- as the reference
 * does, and as torch does unless told otherwise -- needs no synchronisation of
 * its own.  A caller that produces or consumes on another stream passes that
 * stream here; the context then runs on it and the fences are skipped. */
int sift3d_set_stream(sift3d_ctx *ctx, void *hip_stream);
int sift3d_sync(sift3d_ctx *ctx);
void sift3d_free(void *p);

/* ---- Gaussian taps (host) ---------------------------------------------------
 * calculate_gaussian_filter_size + generate_gaussian_filter1d
 * (R/src_common/GaussianMask.cpp:12-57,241-265) and the normalisation of
 * gb3d_blur3d_interleave (R/src_common/GaussBlur3D.cpp:1190-1201).
 * Returns the (odd) tap count, or a negative status; taps must hold 129 floats. */
int sift3d_gauss_taps(float sigma, float min_value, float *taps);
/* Which build of the reference the taps follow.  GaussianMask.cpp calls exp() on a float.  A current g++ (libstdc++ >= 6:
 * <math.h> brings the C++ overloads) makes that expf() and forms the tap's product with the scale in float -- the reference
 * as it compiles today, the default here and what the oracle follows.  The toolchain of the CPU binary the reference
 * repository ships (R/bin/Linux/featExtract, GCC 5.4) made it the C exp(double), with the product formed in double and
 * rounded once (its disassembly at 0x451c87, 0x451d0e, 0x4523b7-0x4523ca).  The two differ by one or two units in the last
 * place of a tap, which every later stage carries into the last printed digits of a record.  With SIFT3D_LIBM_GCC5 the
 * extraction reproduces that binary's .key files byte for byte (tests/test_gpu_parity.py::
 * test_cli_reproduces_the_shipped_binary).  Process-wide (sift3d_gauss_taps takes no context); contexts keep the taps of
 * their patch filters from creation, so choose before sift3d_create.  Returns the previous setting, or SIFT3D_ERR_ARG. */
#define SIFT3D_LIBM_CURRENT 0
#define SIFT3D_LIBM_GCC5 1
int sift3d_set_libm_variant(int which);
int sift3d_get_libm_variant(void);

/* ---- operator level: the reference's four accelerator entry points ---------
 * Host-pointer forms copy in, run on the device and copy the result back
 * (what every reference wrapper does); *_dev forms take device pointers and
 * stay asynchronous on the context's stream. */

/* gb3d_blur3d -> blur_3d_simpleborders_CUDA_Row_Col_Shared_mem
 * (R/cuda_common/SIFT_cuda_Tools.cuh:69-76, called from
 * R/src_common/GaussBlur3D.cpp:1240-1245); the input is NOT clobbered. */
int sift3d_gauss_blur(sift3d_ctx *ctx, const float *in, float *out, int64_t nx, int64_t ny, int64_t nz, float sigma,
                      float min_value);
int sift3d_gauss_blur_dev(sift3d_ctx *ctx, const float *d_in, float *d_out, int64_t nx, int64_t ny, int64_t nz,
                          float sigma, float min_value);
/* Fused form used by the pyramid: out = blur(in), dog = in - out.  d_out may be NULL when only the DoG is wanted
 * (the pyramid does that for its sixth level); d_dog may be NULL. */
int sift3d_gauss_blur_dog_dev(sift3d_ctx *ctx, const float *d_in, float *d_out, float *d_dog, int64_t nx, int64_t ny,
                              int64_t nz, float sigma, float min_value);
/* The same plus what the next octave starts from: d_half = the 2 x 2 x 2 mean of the blurred volume, a dense
 * (nx / 2) x (ny / 2) x (nz / 2) array (fioSubSampleInterpolate, R/src_common/FeatureIO.cpp:1474-1554, called on level 3 of
 * every octave at R/src_common/MultiScale.cpp:409-413).  Where the shape allows (an 11-tap filter -- the pyramid's level 3 --
 * on at least 2^22 voxels, nx a multiple of 8) the blur launch writes it from the planes it holds in registers and
 * *in_one_launch (optional) is 1; otherwise a subsample launch follows and it is 0.  The bytes are the same either way. */
int sift3d_gauss_blur_dog_half_dev(sift3d_ctx *ctx, const float *d_in, float *d_out, float *d_dog, float *d_half, int64_t nx,
                                   int64_t ny, int64_t nz, float sigma, float min_value, int *in_one_launch);
/* The same restricted to the output planes [z_lo, z_hi) of the volume: planes outside the window are not written, the
 * input is read as far as the filter reaches (zeros beyond the volume).  A Z-slab rank filters its two boundary bands
 * with it first, hands them to the halo exchange, and filters the interior while they travel.
 * Only the one-launch form of the blur has a window: sift3d_blur_window_supported says whether this row length and
 * filter take it (rows of whole 16-byte vectors, a plane below 2^29 voxels, at most 17 taps). */
int sift3d_blur_window_supported(int64_t nx, int64_t ny, float sigma, float min_value);
int sift3d_gauss_blur_dog_window_dev(sift3d_ctx *ctx, const float *d_in, float *d_out, float *d_dog, int64_t nx, int64_t ny,
                                     int64_t nz, int64_t z_lo, int64_t z_hi, float sigma, float min_value);
/* fioMultSum_interleave(a, b, out, -1.0f) -> fioCudaMultSum
 * (SIFT_cuda_Tools.cuh:213-217, R/src_common/FeatureIO.cpp:1941-1943) */
int sift3d_dog(sift3d_ctx *ctx, const float *a, const float *b, float *out, int64_t n);
int sift3d_dog_dev(sift3d_ctx *ctx, const float *d_a, const float *d_b, float *d_out, int64_t n);
/* Subsample_interleave -> SubSampleInterpolateCuda (SIFT_cuda_Tools.cuh:202-205,
 * R/src_common/FeatureIO.cpp:1556-1564): out is (nx/2)*(ny/2)*(nz/2). */
int sift3d_subsample2(sift3d_ctx *ctx, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out);
int sift3d_subsample2_dev(sift3d_ctx *ctx, const float *d_in, int64_t nx, int64_t ny, int64_t nz, float *d_out);
/* detectExtrema4D_test_interleave -> detectExtrema4D_test_cuda
 * (SIFT_cuda_Tools.cuh:32-38, R/src_common/MultiScale.cpp:1523-1546): strict
 * extrema of d_cur over its 26 neighbours and centre+26 of d_prev; when d_next
 * is not NULL also centre+26 of d_next (the reference's later validation,
 * MultiScale.cpp:425-453).  Lists come back in raster z,y,x order, minima and
 * maxima separately, as the CPU scan produces them.  *n_min / *n_max always
 * receive the exact counts. */
int sift3d_extrema(sift3d_ctx *ctx, const float *d_prev, const float *d_cur, const float *d_next, int64_t nx, int64_t ny,
                   int64_t nz, sift3d_extremum *minima, int64_t cap_min, int64_t *n_min, sift3d_extremum *maxima,
                   int64_t cap_max, int64_t *n_max);
/* fioDoubleSize (R/src_common/FeatureIO.cpp:2452-2548), out is 2nx*2ny*2nz; and
 * fioSubSample2DCenterPixel (:1670-1714), out is (nx/2)*(ny/2)*(nz/2): the -2+ / -2- options. */
int sift3d_double_size(sift3d_ctx *ctx, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out);
int sift3d_halve_size(sift3d_ctx *ctx, const float *in, int64_t nx, int64_t ny, int64_t nz, float *out);

/* ---- pipeline level: msGeneratePyramidDOG3D_efficient + the descriptor loop --
 * (R/src_common/MultiScale.cpp:236-570, R/featExtract/featExtract.cpp:409,474-505).
 * Everything stays on the device between the upload of the volume and the
 * download of the records. */
int sift3d_set_volume(sift3d_ctx *ctx, const float *vol, int64_t nx, int64_t ny, int64_t nz);
int sift3d_set_volume_dev(sift3d_ctx *ctx, const float *d_vol, int64_t nx, int64_t ny, int64_t nz);
/* The -2+ / -2- options without a round trip through the host (R/featExtract/featExtract.cpp:409-421 calls
 * fioDoubleSize / fioSubSample2DCenterPixel on the host image): upload vol (nx*ny*nz), resize it on the device
 * (resize = +1: doubled to 2nx*2ny*2nz, -1: halved to (nx/2)*(ny/2)*(nz/2), 0: as it is) and make the result the
 * context's volume.  The context must have been created for the larger of the two sizes. */
int sift3d_set_volume_resized(sift3d_ctx *ctx, const float *vol, int64_t nx, int64_t ny, int64_t nz, int resize);
/* The same volume arriving in runs of whole z planes (round 5; the reference reads the whole file and then copies the whole
 * volume, blocking, once per octave: fioCopy, R/src_common/FeatureIO.cpp:1841-1863): sift3d_set_volume_begin names the shape
 * that will arrive and the resize; sift3d_set_volume_planes queues planes [z0, z0 + n) (host memory, nx*ny*n floats) on the
 * context's stream and returns -- every plane exactly once, in any order; the caller keeps a run of planes unchanged until
 * sift3d_set_volume_end, which runs the resize if one was asked for and returns when the volume is resident.  featExtract
 * uploads what it has read while the rest of a .nii.gz is still being inflated. */
int sift3d_set_volume_begin(sift3d_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int resize);
int sift3d_set_volume_planes(sift3d_ctx *ctx, const float *planes, int64_t z0, int64_t n);
int sift3d_set_volume_end(sift3d_ctx *ctx);
/* Optional: make room now for a run that will find about n_extrema validated extrema (the per-keypoint buffers and the
 * pinned record buffers are otherwise made inside the first extraction, once their size is known: about 25 ms of a 512^3
 * extraction that a process which extracts once -- the command line -- can spend beside its file read instead).  Blob fields
 * yield one extremum per 3 000 voxels.  A run that finds more grows the buffers as before; results never depend on it. */
int sift3d_reserve(sift3d_ctx *ctx, int64_t n_extrema);
/* Scale-space + detection only: validated extrema of every octave/level in the
 * reference's order.  *out is malloc'ed (sift3d_free). */
int sift3d_detect(sift3d_ctx *ctx, float initial_image_scale, sift3d_candidate **out, int64_t *n_out);
/* Full extraction.  initial_image_scale: 1, or 0.5 after -2+; size_factor: what
 * featExtract.cpp:423-427 applies to x,y,z,scale (0.5 for -2+, 2 for -2-);
 * eig_thres: 140 in featExtract.cpp:297.  *out is malloc'ed (sift3d_free). */
int sift3d_extract(sift3d_ctx *ctx, float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                   sift3d_feature **out, int64_t *n_out);
/* One z-slice of Gaussian level `level` (0..4) of octave `octave` as the last sift3d_detect / sift3d_extract left it, copied
 * to the host as a dense nx_o * ny_o array (out must hold the octave's plane; *nx_out / *ny_out, which may be NULL, receive
 * its size).  For the reference's debug output ./image.pgm: the middle slice of octave 0's first blurred level
 * (R/src_common/MultiScale.cpp:373-384), which the featExtract command line of this build writes as the reference does. */
int sift3d_get_level_slice(sift3d_ctx *ctx, int octave, int level, int64_t z, float *out, int64_t *nx_out, int64_t *ny_out);
/* Beyond the reference: stop the pyramid after n octaves.  n = 0 restores the reference's only
 * rule -- halve until a dimension is <= 2 (R/src_common/MultiScale.cpp:337,359-360) -- which is also the default; the
 * command line has no such option and never sets it.  Records are ordered octave-major, so a limited run returns
 * exactly the leading records of the unlimited one. */
int sift3d_set_max_octaves(sift3d_ctx *ctx, int n);
/* Implementation choices a caller may override (tests exercise both sides of each; none is needed for normal use, and
 * none changes a result -- every combination returns the same bytes).  No environment variable is read anywhere in the
 * library. */
typedef enum {
    SIFT3D_TUNE_BLUR_FUSED = 0, /* one fused x+y+z+DoG launch per level: 1 where it pays (default), 0 never, 2 wherever the shape allows */
    SIFT3D_TUNE_FUSED_CHUNKS,   /* z chunks of the fused launch: 0 = cost model (default), n >= 1 forced */
    SIFT3D_TUNE_FUSED_ROWS,     /* rows per thread of the fused launch: 0 = by filter and size (default), 1 or 2 forced */
    SIFT3D_TUNE_LAZY_LEVELS,    /* 1 (default): D0, D4 and L5 of an octave are evaluated around the candidates only; 0: stored and
                                 * filtered in full, as the reference's schedule does (MultiScale.cpp:405-413) */
    SIFT3D_TUNE_TINY_OCTAVE,    /* 1 (default): an octave of at most 4096 voxels is built by one workgroup in one launch; 0: level by level */
    SIFT3D_TUNE_SAMPLER_CAP,    /* workgroups of a CU that may sample a patch at a time in the descriptor kernel: 4 (default); 0: no limit */
    SIFT3D_TUNE_KP_CHUNKS,      /* the per-keypoint stage in n chunks, the keypoint kernel of chunk i+1 on one stream beside the
                                 * descriptor kernel of chunk i on another: 0 (default) or 1 = one launch of each, one after the other
                                 * (the overlap does not pay), up to 16; any value but 0 also switches
                                 * SIFT3D_TUNE_SPLIT_TAIL's schedule off */
    SIFT3D_TUNE_BANDS_FIRST,    /* Z-slab drivers: 1 (default) a rank filters the two boundary bands of a level first and the interior
                                 * while they travel to its neighbours; 0: the level in one piece, then the exchange (round 2) */
    SIFT3D_TUNE_HOST_RECORDS,   /* records per candidate the pinned download buffers are first sized for: 5 (default; blob fields yield
                                 * 4.2), 1..12; a run that yields more grows them (sift3d_host_buffer_grows counts) */
    SIFT3D_TUNE_FUSED_TILE,     /* (x, y) tile of the fused launch's two-rows-per-thread mapping: 0 = by measurement (default), 1 = 64 x 32,
                                 * 2 = 128 x 16 */
    SIFT3D_TUNE_FUSED_SUB,      /* 1 (default): the launch that makes level 3 of an octave also writes the next octave's level 0 (the
                                 * 2 x 2 x 2 mean) where the shape allows; 0: a subsample launch of its own, as before round 4 */
    SIFT3D_TUNE_SPLIT_TAIL,     /* 1 (default): the extrema of octaves 0 and 1 are sorted and their keypoint kernel started while the coarser
                                 * octaves are still being built; 0: one sort and one keypoint launch behind the whole pyramid; 2 (tests): as 1 with room
                                 * for eight extrema of the coarser octaves only, so that the fall-back to the schedule of 0 runs */
    SIFT3D_TUNE_DESC_SEGMENT,   /* descriptor kernel: the record list is dealt to the XCDs in contiguous eighths of segments of 8 n records:
                                 * 32 (default: 256 records a segment, all XCDs in the same part of the list at a time); 0: eighths of the
                                 * whole list (rounds 2 - 3); 1: round-robin */
    SIFT3D_TUNE_FUSED_ORDER,    /* fused blur: which workgroup takes which tile.  Workgroup b of a launch goes to XCD b mod 8.  0: by measurement;
                                 * 1: XCD x walks the x-th eighth of the tiles in (x, y, chunk) order -- whole rows of tiles per XCD (rounds 1 - 4);
                                 * 2: workgroup b takes tile b; 3: column strips -- XCD x owns the tiles of column x mod tiles_x, so that a
                                 * tile's y neighbours share its L2 and an XCD always reads the same byte columns of every row */
    SIFT3D_TUNE_FUSED_STAGGER,  /* fused blur, two-rows-per-thread mapping, 7 - 13 taps: the second-dispatched half of a workgroup's wavefronts
                                 * runs half a step behind the first (z pass and stores of a plane at the start of the next step), each
                                 * (half, role) pair of wavefronts running its own straight-line copy of the march.  0: by measurement
                                 * (default: from 11 taps up); 1: off (the kernel of rounds 2 - 5); 2: on */
    SIFT3D_TUNE_COUNT
} sift3d_tuning;
int sift3d_set_tuning(sift3d_ctx *ctx, int knob, int value);
/* how often a run on this context found its pinned record buffers too small and grew them (see SIFT3D_TUNE_HOST_RECORDS) */
int64_t sift3d_host_buffer_grows(const sift3d_ctx *ctx);

/* Same, without the final host copy: *view points at the context's pinned download buffer and stays
 * valid until the next call on this context (or sift3d_destroy).  Do not free it.  The memory is the caller's to read AND
 * to modify in place until then (featExtract applies the -w transform there); the next run overwrites it. */
int sift3d_extract_view(sift3d_ctx *ctx, float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                        const sift3d_feature **view, int64_t *n_out);

/* ---- Z-slab building blocks (multi-GPU) --------------------------------------
 * The reference has no multi-GPU code; this is the surface the Z-slab driver
 * (3d_sift_cuda_amd/zslab.py: one process per GPU, halo exchange with
 * torch.distributed = RCCL over xGMI) needs on top of the *_dev operators.
 * The caller owns the level buffers (device memory): each holds nz_local
 * slices of a level whose whole volume has nz_global slices, starting at
 * global slice z_offset (slab + halos).  Geometry is always computed in
 * whole-volume coordinates, so records are those of the undivided volume. */
typedef struct {
    const float *img;  /* Gaussian level L_k (device), nx*ny*nz_local */
    const float *dogc; /* DoG level k (device), same shape */
    int64_t nx, ny, nz_local, nz_global, z_offset;
    float sigma_h, sigma_c, sigma_l; /* sigmas of levels k-1, k, k+1 (MultiScale.cpp:463) */
    float octave_factor;             /* 2^octave (MultiScale.cpp:531-543) */
} sift3d_level_desc;
/* Forget the extrema collected so far (also restarts the launch log). */
int sift3d_candidates_reset(sift3d_ctx *ctx);
/* Queue one extrema pass over device buffers of nz_local slices and keep the
 * extrema whose local z lies in [z_lo, z_hi) (and in 1..nz_local-2): a slab
 * keeps its own slices, the halo slices are the neighbour's.  level_id =
 * octave*3 + (DoG level - 1) orders the output. Asynchronous. */
int sift3d_extrema_append_dev(sift3d_ctx *ctx, const float *d_prev, const float *d_cur, const float *d_next, int64_t nx,
                              int64_t ny, int64_t nz_local, int level_id, int64_t z_lo, int64_t z_hi);
/* The same pass when a neighbour level is not stored as a DoG volume -- what the single-device pipeline does for the
 * first and last detection level of an octave:
 *   d_prev == NULL: the level below is g_prev_a - g_prev_b, the two Gaussian levels it is the difference of, read at the
 *                   27 positions around an extremum (as the reference's validateDifferencePeak3D reads a level it never
 *                   materialises, R/src_common/MultiScale.cpp:1135-1223);
 *   d_next == NULL: the level above is g_next - blur(g_next, next_sigma), and that blur is evaluated only at the 27
 *                   positions around each extremum that passed everything else, from the (2R+3)^3 block of g_next around
 *                   it -- g_next must therefore be valid R+1 = 9 slices beyond the slices [z_lo, z_hi) (rows and planes
 *                   outside the buffer read as zero, i.e. as the border of the whole volume).
 * sift3d_lazy_levels_supported: 1 when this shape can take that form (rows of whole 16-byte vectors, a plane below
 * 2^29 voxels, the 17-tap filter of the pyramid's last level); otherwise store the levels and use sift3d_extrema_append_dev.
 * A pure function of its arguments. */
int sift3d_lazy_levels_supported(int64_t nx, int64_t ny, int64_t nz_local, float next_sigma);
int sift3d_extrema_append_lazy_dev(sift3d_ctx *ctx, const float *d_prev, const float *g_prev_a, const float *g_prev_b,
                                   const float *d_cur, const float *d_next, const float *g_next, float next_sigma, int64_t nx,
                                   int64_t ny, int64_t nz_local, int level_id, int64_t z_lo, int64_t z_hi);
/* Sort what was collected and return it with whole-volume coordinates (malloc'ed, sift3d_free). */
int sift3d_candidates_dev(sift3d_ctx *ctx, const sift3d_level_desc *levels, int n_levels, sift3d_candidate **out,
                          int64_t *n_out);
/* Sort what was collected and run the per-keypoint stage on it.  levels[id]
 * describes level_id == id.  *view / *group_view (level_id*2 + is_max per
 * record) point at pinned buffers owned by the context, valid until the next
 * call. */
int sift3d_describe_dev(sift3d_ctx *ctx, const sift3d_level_desc *levels, int n_levels, int desc_mode, float eig_thres,
                        float size_factor, const sift3d_feature **view, const int32_t **group_view, int64_t *n_out);

/* sift3d_describe_dev in two halves (round 5), for a caller that places the records of SEVERAL contexts -- the ranks of a Z-slab run,
 * one process each -- in ONE list, so that nothing is gathered or merged afterwards.  A context's records come out sorted by group
 * (level_id*2 + is_max; SIFT3D_GROUPS of them), and the single-device order is, group by group, one run per rank in rank order; so:
 *   sift3d_describe_dev_counts  sorts, runs the keypoint kernel and returns this context's records per group (*group_counts: SIFT3D_GROUPS
 *                               words owned by the context) and their sum;
 *   the caller                  exchanges the counts between the ranks and lays the runs out: shift[g] = (records of groups < g of all ranks)
 *                               + (records of group g of lower ranks) - (this rank's own records of groups < g);
 *   sift3d_describe_dev_place   runs the descriptor kernel, which stores this context's record i (of group g) at list[i + shift[g]], and
 *                               waits for it.  list: host memory the device can write -- e.g. a shared-memory segment that every rank's
 *                               process maps and has passed to sift3d_host_register; the per-record group words stay in the context
 *                               (*group_view).  list == NULL (it turned out too small): the records go to the context's own pinned
 *                               buffers, *own_view, exactly as after sift3d_describe_dev.
 * sift3d_host_register / _unregister: hipHostRegister (portable, mapped) / hipHostUnregister on the caller's pages. */
#define SIFT3D_GROUPS 193
int sift3d_describe_dev_counts(sift3d_ctx *ctx, const sift3d_level_desc *levels, int n_levels, int desc_mode, float eig_thres,
                               float size_factor, const int32_t **group_counts, int64_t *n_records);
int sift3d_describe_dev_place(sift3d_ctx *ctx, sift3d_feature *list, const int32_t *shift, const sift3d_feature **own_view,
                              const int32_t **group_view, int64_t *n_out);
int sift3d_host_register(void *p, int64_t bytes);
int sift3d_host_unregister(void *p);

/* ---- Z-slab extraction from C: one process, several devices ---------------------
 * The whole volume (host memory) is cut into one Z-slab per entry of `devices` (the same partitioning, halo widths and
 * exchange schedule as the per-process driver above), each slab on its own device, halos moved between the devices with
 * hipMemcpyPeerAsync -- xGMI between the GPUs of a node -- queued behind events, so the host never waits inside the pyramid
 * (one host thread per device queues its launches).
 * The records come back merged in the single-GPU order and are the single-GPU records bit for bit.  A device may be listed
 * more than once (the slab logic rehearsed on one GPU).  A volume too thin to shard (a slab must be 32 slices thick) runs
 * whole on devices[0].  SIFT3D_ERR_COMM: a halo copy or its ordering failed; *err (err_len bytes, may be NULL) gets the text.
 * *out is malloc'ed (sift3d_free). */
typedef struct {
    int32_t n_ranks, sharded_octaves; /* n_ranks: the slabs (the octaves below the sharded ones have a rank of their own on devices[0], not counted) */
    int64_t exchanges;            /* halo copies queued on the critical path + deferred batches */
    int64_t halo_bytes_critical;  /* the 8-slice halo every level needs before the next blur (9 of L4), all ranks */
    int64_t halo_bytes_deferred;  /* the rest of the L1..L3 patch halos, copied beside L4 / L5 / extrema */
    int64_t gather_bytes;         /* the first unsharded octave assembled on devices[0] */
    int64_t n_extrema, n_keypoints, n_records;
    double wall_ms;               /* host wall time of the extraction: upload of the slabs, pyramid, per-keypoint stage, download, merge */
    int64_t halo_bytes_hidden;    /* the part of halo_bytes_critical issued bands-first: copied while the receiver filters its interior */
    int32_t transport;            /* what moved the slices between ranks: SIFT3D_TRANSPORT_PEER_COPY or SIFT3D_TRANSPORT_RCCL */
    int32_t transport_fell_back;  /* 1: RCCL was asked for, but a device is listed more than once (not something RCCL ranks can
                                   * be): peer copies were used */
    int32_t rccl_version;         /* ncclGetVersion() of the library that was loaded (0 with peer copies) */
    int32_t comm_sets;            /* RCCL communicator sets in use: 2, or 1 with SIFT3D_ZSLAB_SERIAL_CHANNELS (0 with peer copies) */
    int32_t resident_volume;      /* 1: the input slices were on the devices already (sift3d_zslab_extract_resident): no upload in wall_ms */
    int32_t list_grown;           /* 1: the merged list had to be replaced by a larger one AFTER the slabs' records were in it (the coarse octaves'
                                   * records did not fit behind them): the slabs' part was copied over.  Rare; tests force it */
    double merge_ms;              /* host time spent on the merged order (part of wall_ms).  Round 5: the ranks' descriptor kernels store their
                                   * records straight into their places in ONE pinned list, so this is the per-group offset table and its
                                   * upload (microseconds) plus, when the list has to grow, its allocation -- there is no merge left */
    int64_t halo_bytes_subsample; /* the part of halo_bytes_deferred the next octave waits for: the eight slices of L3 beyond +- 8 that the
                                   * subsample reads (first deferred step); the rest is waited for before the per-keypoint stage only */
    double enqueue_ms;            /* host time from the start of the call until the last rank's pyramid, extrema passes and count request are
                                   * queued (before its first host wait).  One host thread per rank since the second half of round 5 */
} sift3d_zslab_stats;
/* How a block of slices travels from one rank's device to another's (sift3d_zslab_set_tuning(h, SIFT3D_ZSLAB_TRANSPORT, v),
 * sift3d_extract_zslab_over): peer copies -- hipMemcpyPeerAsync on the receiver's stream behind the sender's event, the
 * default -- or RCCL: ncclSend / ncclRecv on the ranks' halo streams inside one ncclGroupStart / ncclGroupEnd per exchange
 * step, one communicator set for the halos a launch waits for and one for the deferred patch halos.  RCCL is loaded at run
 * time (librccl.so.1; sift3d_zslab_set_transport_library names another build, NULL restores the default); a failure to load
 * it or to create the communicators is SIFT3D_ERR_COMM with the reason in err.  NEITHER transport has moved a byte between
 * two GPUs yet (one-GPU development box). */
#define SIFT3D_ZSLAB_TRANSPORT 1000
#define SIFT3D_TRANSPORT_PEER_COPY 0
#define SIFT3D_TRANSPORT_RCCL 1
/* sift3d_zslab_set_tuning(h, SIFT3D_ZSLAB_SERIAL_CHANNELS, 1): RCCL with ONE communicator set -- the deferred patch halos go
 * through the communicators of the per-level halos and queue behind them (RCCL orders a communicator's operations).  Slower
 * by design (the deferred slices in front of the next level's 8); it exists as the fallback to try in the same lease if two
 * communicators per device ever stall each other on real links.  Results are the same bytes. */
#define SIFT3D_ZSLAB_SERIAL_CHANNELS 1001
/* sift3d_zslab_set_tuning(h, SIFT3D_ZSLAB_DUPLICATE_RANKS, 1): a device listed more than once is handed to ncclCommInitAll as
 * it is instead of falling back to peer copies.  Real RCCL refuses such a list (SIFT3D_ERR_COMM); the rehearsal library of
 * tests/rccl_shim (given to sift3d_zslab_set_transport_library) accepts it, which is how the RCCL half of the exchange --
 * group pairing, stream order, the two communicator sets -- runs with 2 .. 8 ranks on a one-GPU box. */
#define SIFT3D_ZSLAB_DUPLICATE_RANKS 1002
/* sift3d_zslab_set_tuning(h, SIFT3D_ZSLAB_POISON_HALO, 1) (tests): every halo slice of L1..L3 that the exchange does NOT fetch is
 * filled with NaN before the per-keypoint stage -- a patch that reached further than the driver's bound would show in the records.
 * 1 + k: the NaN start k slices earlier, inside what was fetched (never inside what the pyramid itself reads): the smallest k that
 * changes a record is the margin the bound has on that volume. */
#define SIFT3D_ZSLAB_POISON_HALO 1003
/* sift3d_zslab_set_tuning(h, SIFT3D_ZSLAB_PATCH_WAIT, v): when a rank waits for the halo slices of L1..L3 that only patches read.
 * 0 (default): before its per-keypoint stage -- they have every later octave to arrive in; 1: at the end of their own octave, with
 * the subsample's slices (rounds 2 - 4).  Same bytes either way. */
#define SIFT3D_ZSLAB_PATCH_WAIT 1004
/* sift3d_zslab_set_tuning(h, SIFT3D_ZSLAB_LIST_ROOM, n) (tests): a newly allocated merged list has room for n records behind the slabs'
 * (the octaves that are not sharded append theirs there); -1 (default): an eighth of the slabs' records + 4096.  0 forces the growth
 * path whenever those octaves have a record.  The handle's list is dropped by the call. */
#define SIFT3D_ZSLAB_LIST_ROOM 1005
void sift3d_zslab_set_transport_library(const char *path);
int sift3d_extract_zslab(const int *devices, int n_devices, const float *vol, int64_t nx, int64_t ny, int64_t nz,
                         float initial_image_scale, int desc_mode, float eig_thres, float size_factor, sift3d_feature **out,
                         int64_t *n_out, sift3d_zslab_stats *stats, char *err, int64_t err_len);
/* sift3d_extract_zslab with the transport named (SIFT3D_TRANSPORT_*); sift3d_extract_zslab uses peer copies */
int sift3d_extract_zslab_over(int transport, const int *devices, int n_devices, const float *vol, int64_t nx, int64_t ny, int64_t nz,
                              float initial_image_scale, int desc_mode, float eig_thres, float size_factor, sift3d_feature **out,
                              int64_t *n_out, sift3d_zslab_stats *stats, char *err, int64_t err_len);
/* The same with the contexts, streams and events of the slabs kept between volumes of one shape: create once, extract any
 * number of volumes (the level buffers of a run come from one arena per slab, sized after the first run), destroy.
 * sift3d_extract_zslab is create + extract + destroy; creating the contexts dominates its wall time. */
typedef struct sift3d_zslab sift3d_zslab;
sift3d_zslab *sift3d_zslab_create(const int *devices, int n_devices, int64_t nx, int64_t ny, int64_t nz, char *err, int64_t err_len);
int sift3d_zslab_extract(sift3d_zslab *h, const float *vol, float initial_image_scale, int desc_mode, float eig_thres,
                         float size_factor, sift3d_feature **out, int64_t *n_out, sift3d_zslab_stats *stats, char *err,
                         int64_t err_len);
void sift3d_zslab_destroy(sift3d_zslab *h);
/* The resident form (round 5) -- what sift3d_set_volume + sift3d_extract_view are to one device: sift3d_zslab_set_volume
 * cuts the host volume into the ranks' input slices (slab +- 16) and uploads them once; sift3d_zslab_extract_resident then
 * runs any number of extractions from HBM.  *view is the handle's own host buffer with the merged records (single-GPU order,
 * single-GPU bytes), valid until the handle's next call; do not free it.  stats->wall_ms then holds no upload. */
/* (Both extraction forms: the ranks' records are placed in the merged order by the descriptor kernels themselves -- a rank's
 * records are sorted by (level, is_max) group already, so once every rank's records per group are known, a 193-word read-back
 * beside the record total each rank waits for anyway, a record's merged position is its own position plus a per-group shift.) */
int sift3d_zslab_set_volume(sift3d_zslab *h, const float *vol, char *err, int64_t err_len);
int sift3d_zslab_extract_resident(sift3d_zslab *h, float initial_image_scale, int desc_mode, float eig_thres, float size_factor,
                                  const sift3d_feature **view, int64_t *n_out, sift3d_zslab_stats *stats, char *err, int64_t err_len);
/* sift3d_set_tuning on every slab's context (and the driver's own use of SIFT3D_TUNE_LAZY_LEVELS) */
int sift3d_zslab_set_tuning(sift3d_zslab *h, int knob, int value);

/* ---- matcher: exact nearest neighbours of 64-component descriptors ------------------
 * The search step of featMatchMultiple (R/feat_common/featMatchUtilities.cpp:1612: flann_find_nearest_neighbors_index over
 * a kd-tree forest built at :1559 with 8 trees, 64 checks -- approximate and randomised; FLANN is not part of
 * /root/reference), done exactly on the matrix cores: squared Euclidean distances from an int8 Gram matrix.
 * db / queries: n x 64 components, each 0..127 (the rank descriptors of a .key file are 0..63).  For every query the k
 * (1..32) nearest database vectors, ascending by (distance, database index): idx and dist2 hold n_q x k entries (-1 /
 * INT32_MAX past the end of a database smaller than k).  A vector that is in both sets finds itself at distance 0, as in
 * the reference, whose vote stage drops the hits inside the query's own image.  repeats > 1 runs the search that many times
 * and reports the mean device time of the runs after the first in *kernel_ms (may be NULL); with repeats == 1 the figure
 * is the one run end to end on the device's clock, including the read-back that validates the bytes.  n_db and n_q at most
 * 2^31 - 4096 (32-bit row indices, the last tile padded). */
int sift3d_knn64(int device, const int8_t *db, int64_t n_db, const int8_t *queries, int64_t n_q, int k, int32_t *idx,
                 int32_t *dist2, int repeats, double *kernel_ms, char *err, int64_t err_len);

/* ---- measurement ------------------------------------------------------------
 * Device time per stage of the last sift3d_detect/sift3d_extract call, from
 * HIP events recorded on the stream the kernels ran on. */
typedef enum {
    SIFT3D_STAGE_BLUR_X = 0, /* separable pass along x */
    SIFT3D_STAGE_BLUR_Y,     /* pass along y */
    SIFT3D_STAGE_BLUR_Z_DOG, /* pass along z with fused DoG store */
    SIFT3D_STAGE_SUBSAMPLE,
    SIFT3D_STAGE_EXTREMA,
    SIFT3D_STAGE_KEYPOINT,   /* refinement, patch, orientation frames */
    SIFT3D_STAGE_DESCRIPTOR,
    SIFT3D_STAGE_BLUR_FUSED, /* x, y, z passes and the DoG store in one kernel */
    SIFT3D_STAGE_OCTAVE_TINY, /* all five levels and DoGs of an octave of at most 4096 voxels in one workgroup */
    SIFT3D_STAGE_COUNT
} sift3d_stage;
typedef struct {
    double ms[SIFT3D_STAGE_COUNT];        /* summed kernel time */
    int64_t launches[SIFT3D_STAGE_COUNT];
    double alg_bytes[SIFT3D_STAGE_COUNT]; /* algorithmic bytes moved */
    int64_t n_octaves, n_extrema, n_keypoints, n_records;
    double total_ms;                      /* first kernel to last, on the stream */
} sift3d_timings;
/* on: 0 off; 1 every launch bracketed by two events (full per-stage breakdown; costs about 1 ms per 512^3 run);
 * 2 only the blur launches on the full-size volume (the dominant kernels: what a benchmark can leave on);
 * 3 as 1, with the extrema passes kept on the main stream instead of running beside the coarser octaves' blurs: slower
 *   overall, but every launch is then timed alone (exclusive kernel times). */
int sift3d_enable_timing(sift3d_ctx *ctx, int on);
int sift3d_get_timings(const sift3d_ctx *ctx, sift3d_timings *t);
/* Per-launch log of the same call (needs timing enabled): one entry per kernel
 * launch in issue order, so a bench can group by kernel instantiation
 * (stage + tap count) and compare with a rocprofv3 kernel trace. */
typedef struct {
    int32_t stage;  /* sift3d_stage */
    int32_t ntaps;  /* Gaussian tap count of a blur launch, else 0 */
    int64_t nvox;   /* voxels (or keypoints / records) the launch covered */
    double alg_bytes;
    double ms;
    double start_ms; /* when the launch began, counted from the first timed launch of the call (a timeline across the streams) */
} sift3d_launch_record;
int sift3d_get_launch_log(const sift3d_ctx *ctx, sift3d_launch_record *out, int64_t cap, int64_t *n);

/* Development hooks (make DEV=1 builds only) and the one hardware self-test the product library carries are declared in
 * include/sift3d_dev.h: they are not part of the boundary a caller of the reference binds. */

#ifdef __cplusplus
}
#endif
#endif
