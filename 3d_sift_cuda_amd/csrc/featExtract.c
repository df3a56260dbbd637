/*
 * featExtract.c -- the reference's command line over the MI355X C-ABI.
 *
 * Mirrors main() of R/featExtract/featExtract.cpp:273-585 (R/ =
 * /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/): same option
 * letters, same usage text and exit codes, same stdout lines, same .key file.
 * -b / -br / -bn are the descriptor switches documented in the reference's
 * README (/root/reference/README.md:26-34).  Differences, all listed in
 * -d<N> allocates and runs on HIP device N (the reference
 * allocates on device 0 and launches on N); without -d the reference runs its
 * CPU code, this build has no CPU path and runs on device 0 (results are
 * those of the CPU path by construction).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "keyfile.h"
#include "nifti_min.h"
#include "sift3d.h"
#include "world.h"

/* SIFT3D_CLI_TIMES=1: wall time of every phase on stderr (the reference prints its "#us" lines the same way) */
static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int print_options(void)
{
    printf("Volumetric local feature extraction v1.1\n");
    printf("Usage: %s [options] <input image> <output features>\n", "featExtract");
    printf("  <input image>: nifti (.nii,.hdr,.nii.gz).\n");
    printf("  <output features>: output file with features.\n");
    printf(" [options]\n");
    printf("  -w         : output feature geometry in world coordinates, NIFTI qto_xyz matrix (default is voxel units).\n");
    printf("  -2+        : double input image size.\n");
    printf("  -2-        : halve input image size.\n");
    printf("  -d[1-9]    : set device id to be used.\n");
    return 0;
}

/* What msGeneratePyramidDOG3D_efficient leaves behind besides its result (R/src_common/MultiScale.cpp): "\n#<microseconds>\n"
 * after the initial blur (:296-302) and after the first blur of every octave (:386-388), "done.\n" at the end of every octave
 * (:558), and ./image.pgm, the middle slice of octave 0's first blurred level scaled to 0..255 (:373-384, output_float in
 * R/src_common/PpImageFloatOutput.cpp:136-166, the 8-bit branch of GenericImage::WriteToFile).  The durations printed here are
 * the device times of the same blurs. */
static void reference_side_effects(sift3d_ctx *ctx, int64_t X, int64_t Y, int64_t Z)
{
    sift3d_timings tm;
    int64_t nlog = 0;
    if (sift3d_get_timings(ctx, &tm) != SIFT3D_OK) return;
    sift3d_get_launch_log(ctx, NULL, 0, &nlog);
    sift3d_launch_record *log = (sift3d_launch_record *)calloc((size_t)(nlog > 0 ? nlog : 1), sizeof *log);
    if (!log) return;
    sift3d_get_launch_log(ctx, log, nlog, &nlog);
    /* blur launches in issue order, grouped into filter calls: one fused launch, or an x, y, z triple; the octave of a
     * call shows in its voxel count; an octave built whole by one workgroup is one call */
    int printed_initial = 0, octaves_seen = 0;
    int64_t last_vox = -1;
    for (int64_t i = 0; i < nlog; i++) {
        const int st = log[i].stage;
        double us = 0;
        int64_t vox = log[i].nvox;
        if (st == SIFT3D_STAGE_BLUR_FUSED || st == SIFT3D_STAGE_OCTAVE_TINY) us = log[i].ms * 1e3;
        else if (st == SIFT3D_STAGE_BLUR_X && i + 2 < nlog && log[i + 1].stage == SIFT3D_STAGE_BLUR_Y && log[i + 2].stage == SIFT3D_STAGE_BLUR_Z_DOG) {
            us = (log[i].ms + log[i + 1].ms + log[i + 2].ms) * 1e3;
            i += 2;
        } else
            continue;
        if (!printed_initial) { /* the initial blur is the first call */
            printf("\n#%lld\n", (long long)us);
            printed_initial = 1;
            continue;
        }
        if (vox != last_vox) { /* the first blur of an octave */
            if (octaves_seen > 0) printf("done.\n");
            printf("\n#%lld\n", (long long)us);
            last_vox = vox;
            octaves_seen++;
        }
    }
    if (octaves_seen > 0) printf("done.\n");
    free(log);
    /* image.pgm */
    float *slice = (float *)malloc(sizeof(float) * (size_t)(X * Y));
    if (slice && sift3d_get_level_slice(ctx, 0, 1, Z / 2, slice, NULL, NULL) == SIFT3D_OK)
        sift3d_write_pgm("image.pgm", slice, (int)Y, (int)X);
    free(slice);
}

/* What the device side does while the main thread reads (or inflates) the image file -- round 5, review item 2: at 512^3 the
 * command line spent 0.25 s bringing up the HIP runtime and the context BEFORE and AFTER 0.1 - 1.8 s of file reading, one after
 * the other.  The thread (1) initialises the runtime (the first HIP call of the process), (2) creates the context, whose
 * size is known from the file's header, (3) uploads the planes the reader has finished, run by run, so that the volume is
 * resident a few milliseconds after the last byte of the file has been read.  Steps 2 and 3 only where they apply: -w / -ws
 * resample the image on the host first (the context's size is not known from the header), and several devices take the
 * whole volume from the host. */
typedef struct {
    int device, want_ctx, want_upload, resize;
    int want_slabs, n_devices, transport; /* several devices, the volume as the file has it: the slab handle (one context per device) */
    const int *devices;
    sift3d_zslab *zs;
    int64_t cx, cy, cz, X, Y, Z;
    const float *data;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int64_t planes_ready; /* planes of data the reader has finished */
    int read_failed;
    /* results */
    int ndev, rc, uploaded;
    sift3d_ctx *ctx;
    char err[512];
    double t_init, t_ctx, t_resident;
} device_job;

static void *device_thread(void *arg)
{
    device_job *j = (device_job *)arg;
    double t0 = now_s();
    j->ndev = sift3d_device_count();
    j->t_init = now_s() - t0;
    if (j->ndev > 0 && j->want_slabs) { /* the contexts of all listed devices, made side by side while the file is read */
        t0 = now_s();
        j->zs = sift3d_zslab_create(j->devices, j->n_devices, j->X, j->Y, j->Z, j->err, sizeof j->err);
        if (j->zs) sift3d_zslab_set_tuning(j->zs, SIFT3D_ZSLAB_TRANSPORT, j->transport);
        j->t_ctx = now_s() - t0;
        return NULL;
    }
    if (j->ndev <= 0 || j->device >= j->ndev || !j->want_ctx) return NULL;
    t0 = now_s();
    j->ctx = sift3d_create(j->device, j->cx, j->cy, j->cz);
    j->t_ctx = now_s() - t0;
    if (!j->ctx) return NULL;
    /* room for the extrema a volume of this size usually has (one per 2 500 voxels of the image AS THE FILE HAS IT is a fifth
     * above blob fields; a volume doubled by -2+ has about the extrema of the original, not eight times as many), made here,
     * beside the read, instead of inside the one extraction this process runs */
    (void)sift3d_reserve(j->ctx, j->X * j->Y * j->Z / 2500 + 256);
    if (!j->want_upload) return NULL;
    j->rc = sift3d_set_volume_begin(j->ctx, j->X, j->Y, j->Z, j->resize);
    int64_t sent = 0;
    while (j->rc == SIFT3D_OK && sent < j->Z) {
        pthread_mutex_lock(&j->mu);
        while (j->planes_ready == sent && !j->read_failed) pthread_cond_wait(&j->cv, &j->mu);
        const int64_t avail = j->planes_ready;
        const int failed = j->read_failed;
        pthread_mutex_unlock(&j->mu);
        if (failed) return NULL;
        j->rc = sift3d_set_volume_planes(j->ctx, j->data + sent * j->X * j->Y, sent, avail - sent);
        sent = avail;
    }
    if (j->rc == SIFT3D_OK) j->rc = sift3d_set_volume_end(j->ctx);
    if (j->rc != SIFT3D_OK) snprintf(j->err, sizeof j->err, "%s", sift3d_last_error(j->ctx));
    else j->uploaded = 1;
    j->t_resident = now_s();
    return NULL;
}

int main(int argc, char **argv)
{
    const double t_main = now_s();
    if (sift3d_abi_version() != SIFT3D_ABI_VERSION) { /* the library writes whole structures through this program's pointers */
        fprintf(stderr, "featExtract: libsift3d_hip.so has ABI version %d, this program was built against %d\n", sift3d_abi_version(), SIFT3D_ABI_VERSION);
        return -1;
    }
    if (argc < 3) {
        print_options();
        return -1;
    }
    int device = -1;
    int devices[64], n_devices = 0; /* -d0,1,2,3: one Z-slab per listed device */
    int transport = SIFT3D_TRANSPORT_PEER_COPY; /* -d0,1,2,3:rccl */
    int arg = 1;
    int resize = 0;
    int desc_mode = SIFT3D_DESC_SIFT;
    int world_mode = 0;
    const float eig_thres = 140;
    while (arg < argc && argv[arg][0] == '-') {
        switch (argv[arg][1]) {
        case '2':
            resize = 1;
            if (argv[arg][2] == '-') resize = -1;
            arg++;
            break;
        case 'd':
            /* (device 0 passes this test whatever the count is, so the runtime is not brought up here for it: that happens
             * beside the file read, in device_thread) */
            if (argv[arg][2] - '0' < 0 || (argv[arg][2] - '0' > 0 && argv[arg][2] - '0' > sift3d_device_count())) {
                printf("Error: unknown device: %d\n", argv[arg][2] - '0');
                print_options();
                return -1;
            }
            device = argv[arg][2] - '0';
            /* beyond the reference: -d0,1,2,3 cuts the volume into one Z-slab per listed device (sift3d_extract_zslab) */
            n_devices = 0;
            for (const char *p = argv[arg] + 2; *p && n_devices < 64; p++) {
                if (*p == ',') continue;
                if (*p == ':') { /* -d0,1,2,3:rccl -- the slabs' halos over RCCL instead of peer copies */
                    if (strcmp(p, ":rccl") == 0) {
                        transport = SIFT3D_TRANSPORT_RCCL;
                        /* (stderr: the usage text and stdout stay the reference's) */
                        fprintf(stderr, "featExtract: the RCCL slab transport is EXPERIMENTAL -- rehearsed on one GPU against a stand-in library only, never "
                                        "run between two GPUs; \":peer\" (the default) moves the halos with peer copies over the same links\n");
                    }
                    else if (strcmp(p, ":peer") != 0) {
                        printf("Error: unknown slab transport: %s\n", p + 1);
                        print_options();
                        return -1;
                    }
                    break;
                }
                if (*p < '0' || *p > '9' || (*p > '0' && *p - '0' >= sift3d_device_count())) {
                    printf("Error: unknown device: %d\n", *p - '0');
                    print_options();
                    return -1;
                }
                devices[n_devices++] = *p - '0';
            }
            arg++;
            break;
        case 'b':
            desc_mode = argv[arg][2] == 'r' ? SIFT3D_DESC_RRIEF : (argv[arg][2] == 'n' ? SIFT3D_DESC_NRRIEF : SIFT3D_DESC_BRIEF);
            arg++;
            break;
        case 'w':
        case 'W':
            /* world coordinates imply isotropic extraction (featExtract.cpp:330-342) */
            world_mode = 1;
            if (argv[arg][2] == 's' || argv[arg][2] == 'S') world_mode = 2;
            arg++;
            break;
        case '-':
            /* beyond the reference (its switch has no '-' case: there "--..." is an unknown argument): --libm=gcc5 makes the
             * Gaussian taps those of the CPU binary the reference repository ships (exp() of a float evaluated by the C
             * exp(double), include/sift3d.h: sift3d_set_libm_variant), so that the .key file is that binary's byte for byte;
             * --libm=current is the default, the reference as a current g++ compiles it */
            if (strcmp(argv[arg], "--libm=gcc5") == 0) sift3d_set_libm_variant(SIFT3D_LIBM_GCC5);
            else if (strcmp(argv[arg], "--libm=current") == 0) sift3d_set_libm_variant(SIFT3D_LIBM_CURRENT);
            else {
                printf("Error: unknown command line argument: %s\n", argv[arg]);
                print_options();
                return -1;
            }
            arg++;
            break;
        default:
            printf("Error: unknown command line argument: %s\n", argv[arg]);
            print_options();
            return -1;
        }
    }
    if (argc - arg < 2) {
        print_options();
        return -1;
    }
    printf("Extracting features: %s\n", argv[arg]);
    const int times = getenv("SIFT3D_CLI_TIMES") != NULL;
    double t0 = now_s(), t1;

    nifti_min_image img;
    nifti_min_stream *in = NULL;
    if (nifti_min_open(argv[arg], &img, &in) < 0) {
        printf("Error: could not read input file: %s\n", argv[arg]);
        return -1;
    }
    if (device < 0) device = 0; /* no CPU path in this build */
    const int multi = n_devices > 1;
    int64_t X = img.nx, Y = img.ny, Z = img.nz;
    int64_t PX = X, PY = Y, PZ = Z; /* processing size */
    float initial_scale = 1.0f;
    if (resize == 1) {
        PX *= 2; PY *= 2; PZ *= 2;
    } else if (resize == -1) {
        PX /= 2; PY /= 2; PZ /= 2;
    }
    /* the device side, beside the read: the runtime; the context when its size is known from the header (not with -w / -ws,
     * which resample first) and wanted (several devices: only for a resize); the upload when one device takes the volume as
     * the file has it */
    device_job job;
    memset(&job, 0, sizeof job);
    job.device = device;
    job.resize = resize;
    job.X = X; job.Y = Y; job.Z = Z;
    job.cx = PX > X ? PX : X; job.cy = PY > Y ? PY : Y; job.cz = PZ > Z ? PZ : Z;
    job.want_ctx = !world_mode && (!multi || resize != 0) && PZ > 1 && PX > 0 && PY > 0;
    job.want_upload = job.want_ctx && !multi && img.nt == 1;
    job.want_slabs = multi && !world_mode && resize == 0 && Z > 1;
    job.devices = devices;
    job.n_devices = n_devices;
    job.transport = transport;
    pthread_mutex_init(&job.mu, NULL);
    pthread_cond_init(&job.cv, NULL);
    const size_t nvox = (size_t)X * (size_t)Y * (size_t)Z * (size_t)img.nt;
    img.data = (float *)malloc(nvox * sizeof(float));
    job.data = img.data;
    pthread_t th;
    const int threaded = img.data && pthread_create(&th, NULL, device_thread, &job) == 0;
    int read_rc = img.data ? 0 : -4;
    {
        /* runs of whole planes, about 32 MB each */
        const size_t plane = (size_t)X * (size_t)Y;
        size_t run = ((size_t)8 << 20) / (plane ? plane : 1);
        if (run < 1) run = 1;
        const size_t planes_total = (size_t)Z * (size_t)img.nt;
        for (size_t z = 0; z < planes_total && read_rc == 0; z += run) {
            const size_t n = planes_total - z < run ? planes_total - z : run;
            read_rc = nifti_min_read_voxels(in, img.data + z * plane, n * plane);
            pthread_mutex_lock(&job.mu);
            if (read_rc == 0) job.planes_ready = (int64_t)(z + n < (size_t)Z ? z + n : (size_t)Z);
            else job.read_failed = 1;
            pthread_cond_signal(&job.cv);
            pthread_mutex_unlock(&job.mu);
        }
        if (!img.data) {
            pthread_mutex_lock(&job.mu);
            job.read_failed = 1;
            pthread_cond_signal(&job.cv);
            pthread_mutex_unlock(&job.mu);
        }
    }
    nifti_min_close(in);
    t1 = now_s();
    if (times) fprintf(stderr, "# read image: %.3f s\n", t1 - t0);
    t0 = t1;
    if (threaded) pthread_join(th, NULL);
    else device_thread(&job); /* no thread: the same steps, one after the other */
    if (read_rc < 0) {
        printf("Error: could not read input file: %s\n", argv[arg]);
        return -1;
    }
    if (world_mode && sift3d_world_make_isotropic(&img) < 0) {
        printf("Error: could not read input file: %s\n", argv[arg]);
        return -1;
    }
    if (job.ndev <= 0 || device >= job.ndev) {
        fprintf(stderr, "Error: no usable HIP device %d.  This build has no CPU fallback: the reference's CPU mode (no -d) is "
                        "oracle/_build/featExtract_oracle in this repository (same options, same .key; test infrastructure, single thread).\n", device);
        return -1;
    }
    if (world_mode) { /* the resampled image is what is processed */
        X = img.nx; Y = img.ny; Z = img.nz;
        PX = X; PY = Y; PZ = Z;
        if (resize == 1) {
            PX *= 2; PY *= 2; PZ *= 2;
        } else if (resize == -1) {
            PX /= 2; PY /= 2; PZ /= 2;
        }
    }
    if (PZ <= 1 || PX <= 0 || PY <= 0) {
        printf("Could not read volume: %s\n", argv[arg]);
        return -1;
    }
    int64_t cx = PX > X ? PX : X, cy = PY > Y ? PY : Y, cz = PZ > Z ? PZ : Z;
    /* several devices: the single-device context is only needed for the -2+ / -2- resize (a volume that needs several
     * GPUs would not fit it otherwise) */
    sift3d_ctx *ctx = job.ctx;
    if (!ctx && (!multi || resize != 0)) {
        double c0 = now_s();
        ctx = sift3d_create(device, cx, cy, cz);
        job.t_ctx = now_s() - c0;
    }
    if (!ctx && (!multi || resize != 0)) {
        printf("Error: could not extract features, insufficient memory.\n");
        return -1;
    }
    t1 = now_s();
    if (times) fprintf(stderr, "# hip runtime: %.3f s, device context: %.3f s (beside the read where the header gives the size); waited %.3f s for them after the read\n", job.t_init, job.t_ctx, t1 - t0);
    t0 = t1;
    /* -2+ / -2-: the resize happens on the device, between the upload and the pyramid */
    if (resize == 1) initial_scale *= 0.5;
    printf("Input image: i=%d j=%d k=%d\n", (int)PX, (int)PY, (int)PZ);

    float size_factor = 1;
    if (resize > 0) size_factor /= 2;
    else if (resize < 0) size_factor *= 2;

    sift3d_feature *feats = NULL;
    int feats_owned = 1; /* malloc'ed by the library (several devices) or a view of the context's download buffer */
    int64_t n = 0;
    int rc = SIFT3D_OK;
    if (multi) {
        /* Z-slabs over the listed devices: the processing volume on the host (resized on the first device if asked),
         * then one slab per device with halos by peer copies */
        float *pv = img.data;
        char zerr[512] = "";
        sift3d_zslab_stats zst;
        if (resize != 0) {
            pv = (float *)malloc(sizeof(float) * (size_t)(PX * PY * PZ));
            if (!pv) rc = SIFT3D_ERR_MEMORY;
            else rc = resize > 0 ? sift3d_double_size(ctx, img.data, X, Y, Z, pv) : sift3d_halve_size(ctx, img.data, X, Y, Z, pv);
            if (rc != SIFT3D_OK) snprintf(zerr, sizeof zerr, "%s", ctx ? sift3d_last_error(ctx) : "out of memory");
            sift3d_destroy(ctx);
            ctx = NULL;
        }
        t1 = now_s();
        if (times) fprintf(stderr, "# resize: %.3f s\n", t1 - t0);
        t0 = t1;
        if (rc == SIFT3D_OK && job.zs) /* the handle the device thread made beside the read (left to the process's end, like the context) */
            rc = sift3d_zslab_extract(job.zs, pv, initial_scale, desc_mode, eig_thres, size_factor, &feats, &n, &zst, zerr, sizeof zerr);
        else if (rc == SIFT3D_OK)
            rc = sift3d_extract_zslab_over(transport, devices, n_devices, pv, PX, PY, PZ, initial_scale, desc_mode, eig_thres, size_factor,
                                           &feats, &n, &zst, zerr, sizeof zerr);
        if (pv != img.data) free(pv);
        if (rc != SIFT3D_OK) {
            fprintf(stderr, "sift3d: %s\n", zerr);
            printf("Error: could not extract features, insufficient memory.\n");
            return -1;
        }
        if (times)
            fprintf(stderr, "# z-slabs: %d ranks, %d sharded octaves, %lld halo transfers over %s%s, %.1f MB on the critical path, %.1f MB deferred, %.1f MB gathered\n",
                    (int)zst.n_ranks, (int)zst.sharded_octaves, (long long)zst.exchanges, zst.transport == SIFT3D_TRANSPORT_RCCL ? "RCCL" : "peer copies",
                    zst.transport_fell_back ? " (RCCL asked for, but a device is listed twice)" : "", zst.halo_bytes_critical / 1e6,
                    zst.halo_bytes_deferred / 1e6, zst.gather_bytes / 1e6);
    } else {
        if (job.uploaded) rc = SIFT3D_OK; /* the planes went up as they were read */
        else if (job.want_upload && job.rc != SIFT3D_OK) {
            rc = job.rc;
            fprintf(stderr, "sift3d: %s\n", job.err);
        } else rc = sift3d_set_volume_resized(ctx, img.data, X, Y, Z, resize);
        t1 = now_s();
        if (times) fprintf(stderr, "# upload: %.3f s%s\n", t1 - t0, job.uploaded ? " (the planes were uploaded while the file was read)" : "");
        t0 = t1;
        if (rc == SIFT3D_OK) rc = sift3d_enable_timing(ctx, 1); /* the reference prints how long its first blurs took */
        /* the records where the descriptor kernel stored them (pinned host memory of the context, ours until the context goes):
         * a copy of 65 MB into fresh pages costs more than the extraction */
        if (rc == SIFT3D_OK) {
            const sift3d_feature *view = NULL;
            rc = sift3d_extract_view(ctx, initial_scale, desc_mode, eig_thres, size_factor, &view, &n);
            feats = (sift3d_feature *)view;
            feats_owned = 0;
        }
        if (rc != SIFT3D_OK) {
            fprintf(stderr, "sift3d: %s\n", sift3d_last_error(ctx));
            printf("Error: could not extract features, insufficient memory.\n");
            return -1;
        }
        reference_side_effects(ctx, PX, PY, PZ);
    }

    t1 = now_s();
    if (times) fprintf(stderr, "# extraction: %.3f s (%lld records)\n", t1 - t0, (long long)n);
    t0 = t1;

    /* %f of a float is at most 47 characters (-3.4e38): twelve matrix entries plus the label need < 700 bytes */
    char c1[200], c2[256], c3[1024];
    snprintf(c1, sizeof c1, "Extraction Voxel Resolution (ijk) : %d %d %d", (int)PX, (int)PY, (int)PZ);
    snprintf(c2, sizeof c2, "Extraction Voxel Size (mm)  (ijk) : %f %f %f", 1.0f * img.dx, 1.0f * img.dy, 1.0f * img.dz);
    if (world_mode) {
        /* featExtract.cpp:447-458, 548-564 */
        float(*m)[4] = img.qto_xyz;
        const char *name = "qto_xyz";
        if (world_mode == 2) {
            if (img.sform_code > 0) {
                m = img.sto_xyz;
                name = "sto_xyz";
            } else {
                printf("Error: sform_code <= 0, output to qto_xyz instead of sto_xyz");
                name = "sto_xyz"; /* the reference keeps the sto_xyz label while using qto_xyz */
            }
        }
        sift3d_world_transform(feats, n, m);
        snprintf(c3, sizeof c3, "Feature Coordinate Space: millimeters (%s) : %f %f %f %f %f %f %f %f %f %f %f %f 0.0 0.0 0.0 1.0", name,
                1.0f * m[0][0], 1.0f * m[0][1], 1.0f * m[0][2], 1.0f * m[0][3], 1.0f * m[1][0], 1.0f * m[1][1], 1.0f * m[1][2],
                1.0f * m[1][3], 1.0f * m[2][0], 1.0f * m[2][1], 1.0f * m[2][2], 1.0f * m[2][3]);
    } else
        snprintf(c3, sizeof c3, "Feature Coordinate Space: voxels: 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0");
    const char *cm[3] = {c1, c2, c3};
    if (sift3d_write_key(argv[arg + 1], feats, n, eig_thres, 3, cm) != 0) {
        fprintf(stderr, "Error: could not write %s\n", argv[arg + 1]);
        return -1;
    }
    t1 = now_s();
    if (times) fprintf(stderr, "# write features: %.3f s\n", t1 - t0);
    printf("\nDone.\n");
    fflush(stdout);
    t0 = t1;
    /* The .key file is closed and complete.  Releasing 7 GB of device memory buffer by buffer, unpinning the download buffers
     * and running the HIP runtime's exit handlers took 0.08 + 0.05 s of a 0.45 s run at 512^3: a process that is about to end
     * leaves that to the kernel driver, which reclaims everything the process held in one go.  SIFT3D_CLI_CLEAN_EXIT=1 keeps
     * the orderly teardown (leak checkers, profilers that flush at exit). */
    if (getenv("SIFT3D_CLI_CLEAN_EXIT") == NULL) {
        if (times) fprintf(stderr, "# teardown: %.3f s\n# main: %.3f s\n", 0.0, now_s() - t_main);
        fflush(NULL);
        _exit(0);
    }
    if (feats_owned) sift3d_free(feats);
    free(img.data);
    sift3d_destroy(ctx);
    sift3d_zslab_destroy(job.zs);
    t1 = now_s();
    if (times) fprintf(stderr, "# teardown: %.3f s\n# main: %.3f s\n", t1 - t0, t1 - t_main);
    return 0;
}
