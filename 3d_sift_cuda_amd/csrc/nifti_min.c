/* nifti_min.c -- see nifti_min.h.  Written against the public NIfTI-1 header
 * layout (348-byte struct, nifti1.h field offsets); no niftilib code. */
#include "nifti_min.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

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
};

void nifti_min_close(nifti_min_stream *s)
{
    if (!s) return;
    if (s->df) gzclose(s->df);
    free(s->raw);
    free(s);
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
    if (single) {
        long off = (long)vox_offset;
        if (off < 352) off = 352;
        if (gzseek(f, off, SEEK_SET) < 0) {
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
        free(ip);
        if (!df) return -2;
        gzbuffer(df, 1u << 20);
        if (vox_offset > 0) gzseek(df, (long)vox_offset, SEEK_SET);
    }
    nifti_min_stream *s = (nifti_min_stream *)calloc(1, sizeof *s);
    if (!s) {
        gzclose(df);
        return -4;
    }
    s->df = df;
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
    switch (s->datatype) {
    case 2: for (size_t i = 0; i < nvox; i++) o[i] = (float)((unsigned char *)raw)[i]; break;
    case 256: for (size_t i = 0; i < nvox; i++) o[i] = (float)((signed char *)raw)[i]; break; /* DT_INT8: plain char may be unsigned */
    case 512: for (size_t i = 0; i < nvox; i++) o[i] = (float)((unsigned short *)raw)[i]; break;
    case 4: for (size_t i = 0; i < nvox; i++) o[i] = (float)((short *)raw)[i]; break;
    case 768: for (size_t i = 0; i < nvox; i++) o[i] = (float)((unsigned int *)raw)[i]; break;
    case 8: for (size_t i = 0; i < nvox; i++) o[i] = (float)((int *)raw)[i]; break;
    case 16: break; /* already in place */
    case 64: for (size_t i = 0; i < nvox; i++) o[i] = (float)((double *)raw)[i]; break;
    }
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
