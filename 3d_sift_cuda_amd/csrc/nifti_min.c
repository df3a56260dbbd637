/* nifti_min.c -- see nifti_min.h.  Written against the public NIfTI-1 header
 * layout (348-byte struct, nifti1.h field offsets); no niftilib code. */
#include "nifti_min.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void swap_bytes(void *p, size_t elem, size_t n)
{
    unsigned char *b = (unsigned char *)p;
    for (size_t i = 0; i < n; i++, b += elem)
        for (size_t j = 0; j < elem / 2; j++) {
            unsigned char t = b[j];
            b[j] = b[elem - 1 - j];
            b[elem - 1 - j] = t;
        }
}

static int16_t rd16(const unsigned char *h, int off, int sw)
{
    int16_t v;
    memcpy(&v, h + off, 2);
    if (sw) swap_bytes(&v, 2, 1);
    return v;
}
static int32_t rd32(const unsigned char *h, int off, int sw)
{
    int32_t v;
    memcpy(&v, h + off, 4);
    if (sw) swap_bytes(&v, 4, 1);
    return v;
}
static float rdf(const unsigned char *h, int off, int sw)
{
    float v;
    memcpy(&v, h + off, 4);
    if (sw) swap_bytes(&v, 4, 1);
    return v;
}

static int ends_with(const char *s, const char *suf)
{
    size_t n = strlen(s), m = strlen(suf);
    return n >= m && strcmp(s + n - m, suf) == 0;
}

/* qform quaternion -> 4x4 (NIfTI-1 standard, method 2) */
static void quatern_to_mat(float qb, float qc, float qd, float qx, float qy, float qz, float dx, float dy, float dz,
                           float qfac, float R[4][4])
{
    double a, b = qb, c = qc, d = qd, xd, yd, zd;
    R[3][0] = R[3][1] = R[3][2] = 0.0f;
    R[3][3] = 1.0f;
    a = 1.0l - (b * b + c * c + d * d);
    if (a < 1.e-7l) {
        a = 1.0l / sqrt(b * b + c * c + d * d);
        b *= a; c *= a; d *= a;
        a = 0.0l;
    } else {
        a = sqrt(a);
    }
    xd = (dx > 0.0) ? dx : 1.0l;
    yd = (dy > 0.0) ? dy : 1.0l;
    zd = (dz > 0.0) ? dz : 1.0l;
    if (qfac < 0.0) zd = -zd;
    R[0][0] = (float)((a * a + b * b - c * c - d * d) * xd);
    R[0][1] = (float)(2.0l * (b * c - a * d) * yd);
    R[0][2] = (float)(2.0l * (b * d + a * c) * zd);
    R[1][0] = (float)(2.0l * (b * c + a * d) * xd);
    R[1][1] = (float)((a * a + c * c - b * b - d * d) * yd);
    R[1][2] = (float)(2.0l * (c * d - a * b) * zd);
    R[2][0] = (float)(2.0l * (b * d - a * c) * xd);
    R[2][1] = (float)(2.0l * (c * d + a * b) * yd);
    R[2][2] = (float)((a * a + d * d - c * c - b * b) * zd);
    R[0][3] = qx; R[1][3] = qy; R[2][3] = qz;
}

static size_t dtype_size(int dt)
{
    switch (dt) {
    case 2: case 256: return 1;
    case 4: case 512: return 2;
    case 8: case 768: case 16: return 4;
    case 64: return 8;
    default: return 0;
    }
}

struct nifti_min_stream {
    gzFile df;
    int datatype, sw;
    size_t es, left; /* bytes per stored voxel; voxels not yet read */
    unsigned char *raw;
    size_t raw_cap;
    unsigned char *mem; /* the whole data file inflated in one go (fast_inflate below), read from mem_pos on; or NULL: df is read */
    size_t mem_pos, mem_n;
};

void nifti_min_close(nifti_min_stream *s)
{
    if (!s) return;
    if (s->df) gzclose(s->df);
    free(s->raw);
    free(s->mem);
    free(s);
}

/* A gzip'ed data file inflated in one call by libdeflate where the system has it (round 5).  zlib's streaming inflate is what
 * the reference reads through (znzlib) and what this reader falls back to; it delivers 0.3 GB/s of voxels, which made
 * `featExtract in.nii.gz out.key` at 512^3 1.5 s of inflate around 0.2 s of everything else.  libdeflate's whole-buffer inflate
 * is 2 - 3 x as fast.  The library is looked for at run time (dlopen "libdeflate.so.0": no header, no link dependency); the
 * whole file must fit in memory twice over, so this path is taken for files that inflate to at most 4 GiB (which also keeps
 * the trailer's size-mod-2^32 field unambiguous), must be ONE gzip member that inflates to exactly what its trailer says, and
 * anything unexpected -- no library, a short file, another size -- leaves the file to zlib as before.  Same bytes either way
 * (tests/test_abi_and_host.py reads every file shape through both). */
typedef void *(*ld_alloc_fn)(void);
typedef int (*ld_gunzip_fn)(void *, const void *, size_t, void *, size_t, size_t *, size_t *);
typedef void (*ld_free_fn)(void *);
static int g_fast_inflate = 1; /* nifti_min_fast_inflate(0): zlib only (tests) */
static int g_fast_taken;       /* files inflated by libdeflate so far (tests, SIFT3D_CLI_TIMES) */
void nifti_min_fast_inflate(int on) { g_fast_inflate = on; }
int nifti_min_fast_inflate_count(void) { return g_fast_taken; }

static int fast_inflate(const char *path, size_t need, unsigned char **out, size_t *out_n)
{
    static void *lib;
    static ld_alloc_fn ld_alloc;
    static ld_gunzip_fn ld_gunzip;
    static ld_free_fn ld_free;
    static int tried;
    if (!g_fast_inflate || need == 0 || need > ((size_t)1 << 32) - 1) return -1;
    if (!tried) { /* (a second thread here at the same moment would load the library twice: harmless) */
        void *l = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (l) {
            ld_alloc = (ld_alloc_fn)dlsym(l, "libdeflate_alloc_decompressor");
            ld_gunzip = (ld_gunzip_fn)dlsym(l, "libdeflate_gzip_decompress_ex");
            ld_free = (ld_free_fn)dlsym(l, "libdeflate_free_decompressor");
            if (ld_alloc && ld_gunzip && ld_free) lib = l;
        }
        __atomic_store_n(&tried, 1, __ATOMIC_RELEASE);
    }
    if (!lib) return -1;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    int rc = -1;
    unsigned char *in = MAP_FAILED, *o = 0;
    void *d = 0;
    size_t n = 0;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 18 + 8 || (uint64_t)sb.st_size > ((uint64_t)1 << 33)) goto end;
    n = (size_t)sb.st_size;
    in = (unsigned char *)mmap(0, n, PROT_READ, MAP_PRIVATE, fd, 0); /* the page cache's own pages: no copy of the compressed file */
    if (in == MAP_FAILED) goto end;
    if (in[0] != 0x1f || in[1] != 0x8b) goto end; /* not gzip: gzopen reads such files as they are */
    {
        const size_t isize = (size_t)in[n - 4] | (size_t)in[n - 3] << 8 | (size_t)in[n - 2] << 16 | (size_t)in[n - 1] << 24;
        if (isize < need) goto end; /* shorter than the voxels the header promises (or several members): zlib's error path decides */
        o = (unsigned char *)malloc(isize);
        d = ld_alloc();
        if (!o || !d) goto end;
        size_t used = 0, got = 0;
        if (ld_gunzip(d, in, n, o, isize, &used, &got) != 0 || got != isize || used != n) goto end;
        *out = o;
        *out_n = got;
        o = 0;
        rc = 0;
        __atomic_fetch_add(&g_fast_taken, 1, __ATOMIC_RELAXED);
    }
end:
    if (d) ld_free(d);
    if (in != MAP_FAILED) munmap(in, n);
    free(o);
    close(fd);
    return rc;
}

int nifti_min_open(const char *path, nifti_min_image *img, nifti_min_stream **out)
{
    unsigned char h[348];
    memset(img, 0, sizeof(*img));
    *out = 0;
    gzFile f = gzopen(path, "rb"); /* gzopen reads plain files transparently */
    if (!f) return -1;
    gzbuffer(f, 1u << 20);
    if (gzread(f, h, 348) != 348) {
        gzclose(f);
        return -1;
    }
    int sw = 0;
    int32_t sizeof_hdr = rd32(h, 0, 0);
    if (sizeof_hdr != 348) {
        sw = 1;
        if (rd32(h, 0, 1) != 348) {
            gzclose(f);
            return -1;
        }
    }
    int ndim = rd16(h, 40, sw);
    if (ndim < 1 || ndim > 7) {
        gzclose(f);
        return -1;
    }
    img->nx = rd16(h, 42, sw);
    img->ny = ndim >= 2 ? rd16(h, 44, sw) : 1;
    img->nz = ndim >= 3 ? rd16(h, 46, sw) : 1;
    img->nt = ndim >= 4 ? rd16(h, 48, sw) : 1;
    if (img->nx < 1) { /* a row length of zero or less is not an image (ny, nz, nt of unused dimensions default to 1) */
        gzclose(f);
        return -1;
    }
    if (img->nt < 1) img->nt = 1;
    if (img->ny < 1) img->ny = 1;
    if (img->nz < 1) img->nz = 1;
    img->datatype = rd16(h, 70, sw);
    float qfac = rdf(h, 76, sw);
    img->dx = rdf(h, 80, sw);
    img->dy = rdf(h, 84, sw);
    img->dz = rdf(h, 88, sw);
    float vox_offset = rdf(h, 108, sw);
    img->qform_code = rd16(h, 252, sw);
    img->sform_code = rd16(h, 254, sw);
    int is_nifti = (h[344] == 'n' && (h[345] == 'i' || h[345] == '+') && h[346] >= '1' && h[346] <= '9' && h[347] == 0);
    int single = is_nifti && h[345] == '+';
    if (!is_nifti) img->qform_code = img->sform_code = 0;

    memset(img->qto_xyz, 0, sizeof(img->qto_xyz));
    memset(img->sto_xyz, 0, sizeof(img->sto_xyz));
    if (img->qform_code > 0) {
        quatern_to_mat(rdf(h, 256, sw), rdf(h, 260, sw), rdf(h, 264, sw), rdf(h, 268, sw), rdf(h, 272, sw), rdf(h, 276, sw),
                       img->dx, img->dy, img->dz, (qfac < 0.0f) ? -1.0f : 1.0f, img->qto_xyz);
    } else { /* method 1: scaling only */
        img->qto_xyz[0][0] = img->dx;
        img->qto_xyz[1][1] = img->dy;
        img->qto_xyz[2][2] = img->dz;
        img->qto_xyz[3][3] = 1.0f;
    }
    if (img->sform_code > 0) {
        for (int j = 0; j < 4; j++) {
            img->sto_xyz[0][j] = rdf(h, 280 + 4 * j, sw);
            img->sto_xyz[1][j] = rdf(h, 296 + 4 * j, sw);
            img->sto_xyz[2][j] = rdf(h, 312 + 4 * j, sw);
        }
        img->sto_xyz[3][3] = 1.0f;
    }

    size_t es = dtype_size(img->datatype);
    if (es == 0) {
        gzclose(f);
        return -3;
    }
    size_t nvox = (size_t)img->nx * (size_t)img->ny * (size_t)img->nz * (size_t)img->nt;
    if (nvox == 0) {
        gzclose(f);
        return -2;
    }
    gzFile df = f;
    unsigned char *mem = 0;
    size_t mem_n = 0, mem_pos = 0;
    if (single) {
        long off = (long)vox_offset;
        if (off < 352) off = 352;
        if (!gzdirect(f) && fast_inflate(path, (size_t)off + nvox * es, &mem, &mem_n) == 0) {
            mem_pos = (size_t)off; /* the voxels are in memory: nothing more is read through zlib */
            gzclose(f);
            df = 0;
        } else if (gzseek(f, off, SEEK_SET) < 0) {
            gzclose(f);
            return -2;
        }
    } else { /* .hdr + .img pair */
        gzclose(f);
        char *ip = (char *)malloc(strlen(path) + 8);
        strcpy(ip, path);
        char *dot = strstr(ip, ".hdr");
        if (!dot) {
            free(ip);
            return -2;
        }
        memcpy(dot, ".img", 4);
        df = gzopen(ip, "rb");
        if (!df && !ends_with(ip, ".gz")) {
            strcat(ip, ".gz");
            df = gzopen(ip, "rb");
        }
        if (!df) {
            free(ip);
            return -2;
        }
        gzbuffer(df, 1u << 20); /* before anything reads or inspects the stream (gzdirect allocates zlib's buffers): later it is refused */
        const size_t off = vox_offset > 0 ? (size_t)(long)vox_offset : 0;
        if (!gzdirect(df) && fast_inflate(ip, off + nvox * es, &mem, &mem_n) == 0) {
            mem_pos = off;
            gzclose(df);
            df = 0;
        } else {
            if (vox_offset > 0) gzseek(df, (long)vox_offset, SEEK_SET);
        }
        free(ip);
    }
    nifti_min_stream *s = (nifti_min_stream *)calloc(1, sizeof *s);
    if (!s) {
        if (df) gzclose(df);
        free(mem);
        return -4;
    }
    s->df = df;
    s->mem = mem;
    s->mem_pos = mem_pos;
    s->mem_n = mem_n;
    s->datatype = img->datatype;
    s->sw = sw;
    s->es = es;
    s->left = nvox;
    *out = s;
    return 0;
}

/* the next nvox voxels of the file, cast to float with a plain C cast (featExtract.cpp:18-77) */
int nifti_min_read_voxels(nifti_min_stream *s, float *dst, size_t nvox)
{
    if (!s || !dst || nvox > s->left) return -2;
    /* float32 voxels are read straight into the result; every other type goes through a raw buffer and a cast */
    const int direct = s->datatype == 16;
    unsigned char *raw = (unsigned char *)dst;
    if (!direct) {
        if (s->raw_cap < nvox * s->es) {
            free(s->raw);
            s->raw = (unsigned char *)malloc(nvox * s->es);
            s->raw_cap = s->raw ? nvox * s->es : 0;
            if (!s->raw) return -4;
        }
        raw = s->raw;
    }
    size_t got = 0, want = nvox * s->es;
    if (s->mem) {
        if (s->mem_pos + want > s->mem_n) return -2;
        memcpy(raw, s->mem + s->mem_pos, want);
        s->mem_pos += want;
        got = want;
    }
    while (got < want) {
        unsigned chunk = (want - got) > (1u << 30) ? (1u << 30) : (unsigned)(want - got);
        int r = gzread(s->df, raw + got, chunk);
        if (r <= 0) break;
        got += (size_t)r;
    }
    if (got != want) return -2;
    s->left -= nvox;
    if (s->sw && s->es > 1) swap_bytes(raw, s->es, nvox);
    float *o = dst;
    /* (the casts of a large run over the host cores -- at most 16 of them: the box may show many more than the process may use) */
#ifdef _OPENMP
    int nt = nvox >= ((size_t)1 << 20) ? omp_get_max_threads() : 1;
    if (nt > 16) nt = 16;
#define CAST_LOOP(T) _Pragma("omp parallel for num_threads(nt) schedule(static)") for (long long i = 0; i < (long long)nvox; i++) o[i] = (float)((const T *)raw)[i]
#else
#define CAST_LOOP(T) for (size_t i = 0; i < nvox; i++) o[i] = (float)((const T *)raw)[i]
#endif
    switch (s->datatype) {
    case 2: CAST_LOOP(unsigned char); break;
    case 256: CAST_LOOP(signed char); break; /* DT_INT8: plain char may be unsigned */
    case 512: CAST_LOOP(unsigned short); break;
    case 4: CAST_LOOP(short); break;
    case 768: CAST_LOOP(unsigned int); break;
    case 8: CAST_LOOP(int); break;
    case 16: break; /* already in place */
    case 64: CAST_LOOP(double); break;
    }
#undef CAST_LOOP
    return 0;
}

int nifti_min_read(const char *path, nifti_min_image *img)
{
    nifti_min_stream *s = 0;
    int rc = nifti_min_open(path, img, &s);
    if (rc < 0) return rc;
    const size_t nvox = (size_t)img->nx * (size_t)img->ny * (size_t)img->nz * (size_t)img->nt;
    img->data = (float *)malloc(nvox * sizeof(float));
    if (!img->data) {
        nifti_min_close(s);
        return -4;
    }
    const size_t step = (size_t)1 << 24; /* 16 M voxels a time: the raw buffer of a cast stays small */
    for (size_t at = 0; at < nvox && rc == 0; at += step) rc = nifti_min_read_voxels(s, img->data + at, nvox - at < step ? nvox - at : step);
    nifti_min_close(s);
    if (rc < 0) {
        free(img->data);
        img->data = 0;
        return rc;
    }
    return 0;
}

int nifti_min_write_f32(const char *path, const float *data, int nx, int ny, int nz, float dx, float dy, float dz)
{
    return nifti_min_write_f32_ex(path, data, nx, ny, nz, dx, dy, dz, 0, 0);
}

int nifti_min_write_f32_ex(const char *path, const float *data, int nx, int ny, int nz, float dx, float dy, float dz,
                           const float *q, const float *srow)
{
    unsigned char h[352];
    memset(h, 0, sizeof(h));
    int32_t sz = 348;
    memcpy(h, &sz, 4);
    int16_t dim[8] = {3, (int16_t)nx, (int16_t)ny, (int16_t)nz, 1, 1, 1, 1};
    memcpy(h + 40, dim, 16);
    int16_t dt = 16, bp = 32;
    memcpy(h + 70, &dt, 2);
    memcpy(h + 72, &bp, 2);
    float pixdim[8] = {1.0f, dx, dy, dz, 1.0f, 1.0f, 1.0f, 1.0f};
    memcpy(h + 76, pixdim, 32);
    float vo = 352.0f, slope = 1.0f;
    memcpy(h + 108, &vo, 4);
    memcpy(h + 112, &slope, 4);
    h[123] = 2; /* xyzt_units: mm */
    if (q) {
        int16_t code = 1;
        memcpy(h + 252, &code, 2);
        memcpy(h + 256, q, 24); /* quatern_b,c,d, qoffset_x,y,z */
        memcpy(h + 76, q + 6, 4); /* pixdim[0] = qfac */
    }
    if (srow) {
        int16_t code = 1;
        memcpy(h + 254, &code, 2);
        memcpy(h + 280, srow, 48);
    }
    memcpy(h + 344, "n+1", 4);
    size_t n = (size_t)nx * ny * nz * 4;
    if (ends_with(path, ".gz")) {
        gzFile f = gzopen(path, "wb1");
        if (!f) return -1;
        gzwrite(f, h, 352);
        size_t put = 0;
        while (put < n) {
            unsigned chunk = (n - put) > (1u << 30) ? (1u << 30) : (unsigned)(n - put);
            if (gzwrite(f, (const unsigned char *)data + put, chunk) <= 0) break;
            put += chunk;
        }
        gzclose(f);
        return put == n ? 0 : -1;
    }
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int ok = fwrite(h, 1, 352, f) == 352 && fwrite(data, 1, n, f) == n;
    fclose(f);
    return ok ? 0 : -1;
}

void nifti_min_free(nifti_min_image *img)
{
    free(img->data);
    img->data = 0;
}
