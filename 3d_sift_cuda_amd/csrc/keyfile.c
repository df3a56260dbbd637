/* keyfile.c -- see keyfile.h */
#include "keyfile.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/types.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int keep(const sift3d_feature *r, float eig_thres)
{
    float es = r->eigs[0] + r->eigs[1] + r->eigs[2];
    float ep = r->eigs[0] * r->eigs[1] * r->eigs[2];
    float esp = es * es * es;
    return (esp < eig_thres * ep || eig_thres < 0);
}

/* The text writer formats 81 numbers per record; at 185 000 records (512^3) fprintf's "%f" alone takes 0.6 s,
 * fifty extractions' worth.  These two produce the same characters. */

/* decimal digits of v, most significant first; returns the end */
static char *put_u64(char *p, uint64_t v)
{
    char tmp[24];
    int k = 0;
    do {
        tmp[k++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}

/* printf("%f", v) for a float: the exact binary value rounded half-to-even at the sixth decimal.  |v| * 10^6 is
 * exact in double (24 mantissa bits times 15625 * 2^6: 38 bits), so rint() of it is that rounding. */
static char *put_f(char *p, float v)
{
    const double d = (double)v;
    if (!(fabs(d) < 4.0e9)) return p + sprintf(p, "%f", d); /* huge, infinite, NaN: rare, let libc spell it */
    if (signbit(v)) *p++ = '-';
    const uint64_t q = (uint64_t)rint(fabs(d) * 1.0e6);
    p = put_u64(p, q / 1000000u);
    *p++ = '.';
    uint32_t fr = (uint32_t)(q % 1000000u);
    for (int k = 5; k >= 0; k--) {
        p[k] = (char)('0' + fr % 10);
        fr /= 10;
    }
    return p + 6;
}

static char *put_int(char *p, long long v)
{
    if (v < 0) {
        *p++ = '-';
        return put_u64(p, (uint64_t)(-v));
    }
    return put_u64(p, (uint64_t)v);
}

/* one record as msFeature3DVectorOutputText prints it: "%f\t" x 16, "%d\t", "%i\t" of (char) x 64, "\n"
 * (MultiScale.h:386-474); a "%f" is at most 48 characters (FLT_MAX), so a record stays below KEY_REC_MAX bytes */
enum { KEY_REC_MAX = 1200, KEY_BLOCK = 2048 /* records formatted per task */ };
static char *put_record(char *p, const sift3d_feature *r)
{
    const float head[4] = {r->x, r->y, r->z, r->scale};
    for (int j = 0; j < 4; j++) { p = put_f(p, head[j]); *p++ = '\t'; }
    for (int j = 0; j < 9; j++) { p = put_f(p, r->ori[j]); *p++ = '\t'; }
    for (int j = 0; j < 3; j++) { p = put_f(p, r->eigs[j]); *p++ = '\t'; }
    p = put_int(p, (int)r->info);
    *p++ = '\t';
    for (int j = 0; j < SIFT3D_DESC_LEN; j++) { p = put_int(p, (char)(r->desc[j])); *p++ = '\t'; }
    *p++ = '\n';
    return p;
}

/* the lengths of the same, without the characters: what lets every block be formatted straight into its place in the file */
static int digits_u64(uint64_t v)
{
    int n = 1;
    while (v >= 10) {
        v /= 10;
        n++;
    }
    return n;
}
static int len_f(float v)
{
    const double d = (double)v;
    if (!(fabs(d) < 4.0e9)) return snprintf(NULL, 0, "%f", d);
    const uint64_t q = (uint64_t)rint(fabs(d) * 1.0e6);
    return (signbit(v) ? 1 : 0) + digits_u64(q / 1000000u) + 7;
}
static int len_int(long long v) { return v < 0 ? 1 + digits_u64((uint64_t)(-v)) : digits_u64((uint64_t)v); }
static size_t len_record(const sift3d_feature *r)
{
    size_t n = (size_t)(len_f(r->x) + len_f(r->y) + len_f(r->z) + len_f(r->scale));
    for (int j = 0; j < 9; j++) n += (size_t)len_f(r->ori[j]);
    for (int j = 0; j < 3; j++) n += (size_t)len_f(r->eigs[j]);
    n += (size_t)len_int((int)r->info);
    for (int j = 0; j < SIFT3D_DESC_LEN; j++) n += (size_t)len_int((char)(r->desc[j]));
    return n + 16 + 1 + SIFT3D_DESC_LEN + 1; /* the tabs and the newline */
}

/* 0: positional writes (the default); 1: a shared mapping of the reserved file where the file system offers one (tests and
 * tools/key_writer_bench.c run both) */
static int g_key_writer_mode = 0;
void sift3d_write_key_mode(int mode) { g_key_writer_mode = mode == 1 ? 1 : 0; }
/* 0: the parallel in-memory reader where the file has the writers' layout (default); 1: fscanf always (tests run both) */
static int g_key_reader_mode = 0;
void sift3d_read_key_mode(int mode) { g_key_reader_mode = mode == 1 ? 1 : 0; }

static int write_all_at(int fd, const char *buf, size_t len, off_t at)
{
    while (len) {
        ssize_t w = pwrite(fd, buf, len, at);
        if (w <= 0) return -1;
        buf += w; len -= (size_t)w; at += w;
    }
    return 0;
}

/* Round 5 (review item 2): at 512^3 the text is 63 MB and 15 million numbers; written by one thread it was the longest phase
 * of `featExtract in.nii out.key` after the file read (0.07 s against 0.01 s of extraction).  The records are cut into blocks
 * of KEY_BLOCK.  Pass 1, in parallel: which records of a block pass the eigenvalue filter and how many characters they will
 * take (the digit counts of the numbers, no formatting).  The block sizes summed are every block's place in the file.  Pass 2,
 * in parallel: every block formatted into a buffer of its thread and written at its place (pwrite).  The other way to get the
 * blocks in -- the file reserved (posix_fallocate) and mapped, every block formatted straight into the mapping -- is kept as
 * mode 1: on the development container (ext4) it is the faster one (0.036 - 0.047 against 0.042 - 0.08 s with 8 threads), on the GPU
 * boxes (overlayfs) page faults on a shared mapping cost twice what the writes do (0.034 against 0.018 s with 16 threads; one
 * thread: 0.096 s; profiles/r05_cli.txt).  The bytes are those of the serial writer: tests/test_oracle_pins.py holds them to the
 * reference's own writer compiled from its header, tests/test_abi_and_host.py to the oracle's fprintf for 1, 3 and 8 threads and
 * both modes. */
int sift3d_write_key(const char *path, const sift3d_feature *recs, int64_t n, float eig_thres, int n_comments,
                     const char *const *comments)
{
    FILE *f = fopen(path, "w+"); /* read-write: a shared mapping that is written needs a descriptor opened for both */
    if (!f) return -1;
    const int64_t nblocks = (n + KEY_BLOCK - 1) / KEY_BLOCK;
    size_t *len = (size_t *)calloc((size_t)(nblocks ? nblocks : 1), sizeof(size_t));
    int64_t *kept = (int64_t *)calloc((size_t)(nblocks ? nblocks : 1), sizeof(int64_t));
    off_t *at = (off_t *)calloc((size_t)(nblocks + 1), sizeof(off_t));
    int bad = !len || !kept || !at;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
    if (nthreads > 16) nthreads = 16;
    if (nblocks < 4) nthreads = 1;
#endif
    int64_t count = 0;
    if (!bad) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
        for (int64_t b = 0; b < nblocks; b++) {
            const int64_t i0 = b * KEY_BLOCK, i1 = i0 + KEY_BLOCK < n ? i0 + KEY_BLOCK : n;
            size_t bytes = 0;
            int64_t k = 0;
            for (int64_t i = i0; i < i1; i++)
                if (keep(&recs[i], eig_thres)) {
                    bytes += len_record(&recs[i]);
                    k++;
                }
            len[b] = bytes;
            kept[b] = k;
        }
        for (int64_t b = 0; b < nblocks; b++) count += kept[b];
    }
    fprintf(f, "# featExtract %s\n", "1.1");
    for (int i = 0; i < n_comments; i++) fprintf(f, "# %s\n", comments[i]);
    fprintf(f, "Features: %d\n", (int)count);
    fprintf(f, "Scale-space location[x y z scale] orientation[o11 o12 o13 o21 o22 o23 o31 o32 o32] 2nd moment "
               "eigenvalues[e1 e2 e3] info flag[i1] descriptor[d1 .. d64]\n");
    if (fflush(f) != 0) bad = 1;
    const off_t head = ftello(f);
    if (head < 0) bad = 1;
    if (!bad) {
        size_t biggest = 0;
        at[0] = head;
        for (int64_t b = 0; b < nblocks; b++) {
            at[b + 1] = at[b] + (off_t)len[b];
            if (len[b] > biggest) biggest = len[b];
        }
        const int fd = fileno(f);
        const off_t total = at[nblocks];
        char *map = (char *)MAP_FAILED;
        if (nthreads > 1 && total > head && g_key_writer_mode == 1 && posix_fallocate(fd, 0, total) == 0)
            map = (char *)mmap(NULL, (size_t)total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        int wbad = 0;
#pragma omp parallel num_threads(nthreads) reduction(| : wbad)
        {
            char *buf = map == (char *)MAP_FAILED ? (char *)malloc(biggest + KEY_REC_MAX) : NULL;
            if (map == (char *)MAP_FAILED && !buf) wbad = 1;
#pragma omp for schedule(dynamic, 1)
            for (int64_t b = 0; b < nblocks; b++) {
                if (!len[b] || wbad) continue;
                const int64_t i0 = b * KEY_BLOCK, i1 = i0 + KEY_BLOCK < n ? i0 + KEY_BLOCK : n;
                char *const dst = buf ? buf : map + at[b];
                char *p = dst;
                for (int64_t i = i0; i < i1; i++)
                    if (keep(&recs[i], eig_thres)) p = put_record(p, &recs[i]);
                if ((size_t)(p - dst) != len[b]) wbad = 1; /* the two passes disagree: never, but then the file is wrong */
                else if (buf) wbad |= write_all_at(fd, buf, len[b], at[b]) != 0;
            }
            free(buf);
        }
        bad |= wbad;
        if (map != (char *)MAP_FAILED && munmap(map, (size_t)total) != 0) bad = 1;
    }
    free(len);
    free(kept);
    free(at);
    if (fclose(f) != 0) bad = 1;
    return bad ? -1 : 0;
}

int sift3d_write_key_bin(const char *path, const sift3d_feature *recs, int64_t n, float eig_thres)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int count = 0;
    for (int64_t i = 0; i < n; i++)
        if (keep(&recs[i], eig_thres)) count++;
    fprintf(f, "# featExtract %s\n", "1.1");
    fprintf(f, "Features: %d\n", count);
    for (int64_t i = 0; i < n; i++) {
        const sift3d_feature *r = &recs[i];
        if (!keep(r, eig_thres)) continue;
        fwrite(&r->x, sizeof(float), 1, f);
        fwrite(&r->y, sizeof(float), 1, f);
        fwrite(&r->z, sizeof(float), 1, f);
        fwrite(&r->scale, sizeof(float), 1, f);
        fwrite(r->ori, sizeof(float), 9, f);
        fwrite(r->eigs, sizeof(float), 3, f);
        fwrite(&r->info, sizeof(unsigned int), 1, f);
        unsigned char pc[SIFT3D_DESC_LEN];
        for (int j = 0; j < SIFT3D_DESC_LEN; j++) pc[j] = (unsigned char)(r->desc[j]);
        fwrite(pc, 1, SIFT3D_DESC_LEN, f);
    }
    fclose(f);
    return 0;
}

/* ---- the text reader ------------------------------------------------------------------------------------------------
 * msFeature3DVectorInputText reads every number with fscanf("%f\t"): 15 million calls for the 63 MB of a 512^3 volume's
 * file, 1.7 s -- five hundred times the matcher's search over the records it delivers.  Round 5: the records of a file in
 * the writers' own layout (one record a line, 81 tab-separated plain decimal numbers) are parsed in parallel from memory;
 * ANYTHING else -- a line with another count of fields, a character that is not part of a plain decimal number (exponents,
 * "inf", "nan", hex floats, a '+'), fewer lines than the header promises -- sends the whole file down the fscanf loop as
 * before, so what is accepted, what is refused and every returned bit are msFeature3DVectorInputText's
 * (tests/test_oracle_pins.py::test_key_reader_against_reference_source holds both paths to the reference's reader).
 *
 * A plain decimal token is converted exactly as strtof would: up to 15 significant digits m and k fraction digits give the
 * double d = m / 10^k correctly rounded (m and 10^k are exact doubles, k <= 22), and (float)d is the correctly rounded
 * float unless d sits on or next to the midpoint of two floats (its low 29 mantissa bits are 0x0FFFFFFF .. 0x10000001), where
 * the second rounding could differ from a single one: those tokens, and longer ones, go through strtof itself. */
static int parse_plain_float(const char *p, const char *end, float *out)
{
    static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *q = p;
    int neg = 0;
    if (q < end && *q == '-') {
        neg = 1;
        q++;
    }
    uint64_t m = 0;
    int nd = 0, k = 0, seen_digit = 0, seen_dot = 0, lead = 1;
    for (; q < end; q++) {
        const char ch = *q;
        if (ch >= '0' && ch <= '9') {
            seen_digit = 1;
            if (lead && ch == '0') { /* leading zeros carry no significance */
                if (seen_dot) k++;
                continue;
            }
            lead = 0;
            m = m * 10u + (uint64_t)(ch - '0');
            nd++;
            if (seen_dot) k++;
            if (nd > 15) return -1; /* beyond what a double holds exactly: strtof */
        } else if (ch == '.' && !seen_dot) {
            seen_dot = 1;
        } else
            return -2; /* not a plain decimal number: the caller gives the file to fscanf */
    }
    if (!seen_digit) return -2;
    if (k > 22) return -1;
    const double d = (double)m / p10[k];
    uint64_t bits;
    memcpy(&bits, &d, sizeof bits);
    const uint32_t low = (uint32_t)(bits & 0x1FFFFFFFu); /* the 29 mantissa bits a float does not keep */
    if (low >= 0x0FFFFFFFu && low <= 0x10000001u) return -1; /* on or beside a midpoint of two floats: one rounding, by strtof */
    const float f = (float)d;
    *out = neg ? -f : f;
    return 0;
}

/* one token through strtof (the conversion fscanf's %f performs); the token must be consumed whole */
static int parse_token_strtof(const char *p, const char *end, float *out)
{
    char tmp[400];
    const size_t len = (size_t)(end - p);
    if (len == 0 || len >= sizeof tmp) return -2;
    memcpy(tmp, p, len);
    tmp[len] = 0;
    char *stop = NULL;
    const float v = strtof(tmp, &stop);
    if (stop != tmp + len) return -2;
    *out = v;
    return 0;
}

/* the fields of one record line [p, end): 16 floats, the info integer, 64 descriptor values, each followed by a tab.
 * 0, or -2 when the line is not in the writers' layout */
static int parse_record_line(const char *p, const char *end, sift3d_feature *r)
{
    float v[16 + 1 + SIFT3D_DESC_LEN];
    int nf = 0;
    while (p < end) {
        const char *t = memchr(p, '\t', (size_t)(end - p));
        if (!t) return -2; /* every field is followed by a tab */
        if (nf >= 16 + 1 + SIFT3D_DESC_LEN) return -2;
        if (nf == 16) { /* "%d": an integer, no point */
            long long iv = 0;
            const char *q = p;
            int neg = 0, digits = 0;
            if (q < t && *q == '-') { neg = 1; q++; }
            for (; q < t; q++) {
                if (*q < '0' || *q > '9') return -2;
                iv = iv * 10 + (*q - '0');
                if (++digits > 10) return -2;
            }
            if (!digits || iv > 2147483647ll + (neg ? 1 : 0)) return -2;
            const int info = (int)(neg ? -iv : iv);
            memcpy(&v[16], &info, sizeof info); /* carried as bits */
        } else {
            int rc = parse_plain_float(p, t, &v[nf]);
            if (rc == -1) rc = parse_token_strtof(p, t, &v[nf]);
            if (rc != 0) return -2;
        }
        nf++;
        p = t + 1;
    }
    if (nf != 16 + 1 + SIFT3D_DESC_LEN) return -2;
    r->x = v[0]; r->y = v[1]; r->z = v[2]; r->scale = v[3];
    memcpy(r->ori, v + 4, sizeof(float) * 9);
    memcpy(r->eigs, v + 13, sizeof(float) * 3);
    memcpy(&r->info, &v[16], sizeof(unsigned int));
    memcpy(r->desc, v + 17, sizeof(float) * SIFT3D_DESC_LEN);
    return 0;
}

/* the records of a file in the writers' layout, from memory and in parallel; 1 = done, 0 = not that layout (nothing returned) */
static int read_records_fast(const char *buf, size_t len, int count, sift3d_feature *r)
{
    /* line starts of the first `count` lines */
    size_t *ls = (size_t *)malloc(sizeof(size_t) * ((size_t)count + 1));
    if (!ls) return 0;
    size_t at = 0;
    int nl = 0;
    while (nl < count && at < len) {
        const char *e = memchr(buf + at, '\n', len - at);
        if (!e) break; /* a last line without its newline: not what the writers produce */
        ls[nl++] = at;
        at = (size_t)(e - buf) + 1;
    }
    ls[nl] = at;
    int ok = nl == count;
    if (ok) {
        int bad = 0;
        int nthreads = 1;
#ifdef _OPENMP
        /* (never the machine's whole core count: a box of the pool shows 256 cores to a 16-CPU share, and 256 threads on 16
         * CPUs made this loop slower than fscanf) */
        nthreads = omp_get_max_threads();
        if (nthreads > 16) nthreads = 16;
        if (count < 4096) nthreads = 1;
#endif
#pragma omp parallel for schedule(static) reduction(| : bad) num_threads(nthreads)
        for (int i = 0; i < count; i++)
            bad |= parse_record_line(buf + ls[i], buf + ls[i + 1] - 1, &r[i]) != 0;
        ok = !bad;
    }
    free(ls);
    return ok;
}

int sift3d_read_key(const char *path, sift3d_feature **recs, int64_t *n)
{
    if (!recs || !n) return -1;
    *recs = NULL;
    *n = 0;
    FILE *f = fopen(path, "rt");
    if (!f) return -1;
    char buff[400];
    buff[0] = '#';
    while (buff[0] == '#') /* read past comments */
        if (!fgets(buff, sizeof(buff), f)) {
            fclose(f);
            return -1;
        }
    int count = 0;
    if (sscanf(buff, "Features: %d\n", &count) <= 0 || count <= 0) {
        fclose(f);
        return -1;
    }
    if (!fgets(buff, sizeof(buff), f) || !strstr(buff, "Scale-space location[x y z scale]")) {
        fclose(f);
        return -1;
    }
    sift3d_feature *r = (sift3d_feature *)calloc((size_t)count, sizeof(sift3d_feature));
    if (!r) {
        fclose(f);
        return -1;
    }
    /* the fast path: the rest of the file in memory, the records parsed in parallel */
    if (g_key_reader_mode == 0) {
        const off_t here = ftello(f);
        if (here >= 0 && fseeko(f, 0, SEEK_END) == 0) {
            const off_t endpos = ftello(f);
            if (endpos > here && fseeko(f, here, SEEK_SET) == 0) {
                const size_t len = (size_t)(endpos - here);
                char *buf = (char *)malloc(len + 1);
                if (buf) {
                    const size_t got = fread(buf, 1, len, f);
                    const int done = got == len && read_records_fast(buf, len, count, r);
                    free(buf);
                    if (done) {
                        fclose(f);
                        *recs = r;
                        *n = count;
                        return 0;
                    }
                    memset(r, 0, sizeof(sift3d_feature) * (size_t)count); /* as calloc left it */
                }
            }
            if (here < 0 || fseeko(f, here, SEEK_SET) != 0) {
                free(r);
                fclose(f);
                return -1;
            }
        }
    }
    for (int i = 0; i < count; i++) {
        int ok = fscanf(f, "%f\t%f\t%f\t%f\t", &r[i].x, &r[i].y, &r[i].z, &r[i].scale) == 4;
        for (int j = 0; j < 9 && ok; j++) ok = fscanf(f, "%f\t", &r[i].ori[j]) == 1;
        for (int j = 0; j < 3 && ok; j++) ok = fscanf(f, "%f\t", &r[i].eigs[j]) == 1;
        int info = 0;
        if (ok) ok = fscanf(f, "%d\t", &info) == 1;
        r[i].info = (unsigned int)info;
        for (int j = 0; j < SIFT3D_DESC_LEN && ok; j++) ok = fscanf(f, "%f\t", &r[i].desc[j]) == 1;
        if (!ok) { /* the reference asserts here */
            free(r);
            fclose(f);
            return -2;
        }
    }
    fclose(f);
    *recs = r;
    *n = count;
    return 0;
}

int sift3d_write_pgm(const char *path, const float *slice, int rows, int cols)
{
    if (!path || !slice || rows < 1 || cols < 1) return -1;
    /* min_max_float (PpImageFloatOutput.cpp:22-63): first strictly greater / smaller value in raster order */
    float lo = slice[0], hi = slice[0];
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = 0; i < n; i++) {
        if (slice[i] > hi) hi = slice[i];
        if (slice[i] < lo) lo = slice[i];
    }
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P5\n%d %d\n%d\n", cols, rows, 255); /* GenericImage::WriteToFile, 8 bits per pixel */
    /* output_float (:136-165): (unsigned char)(((v - min) * 255.0) / (max - min)), the product and quotient in double.  A
     * constant slice divides 0.0 by 0.0f there and casts the NaN -- undefined in C, 0 from x86's cvttsd2si & 0xff; written
     * as 0 here */
    unsigned char *row = (unsigned char *)malloc((size_t)cols);
    if (!row) {
        fclose(f);
        return -1;
    }
    for (int r = 0; r < rows; r++) {
        for (int q = 0; q < cols; q++) {
            const double v = ((slice[(int64_t)r * cols + q] - lo) * 255.0) / (hi - lo);
            row[q] = v == v ? (unsigned char)v : 0;
        }
        fwrite(row, 1, (size_t)cols, f);
    }
    free(row);
    return fclose(f) == 0 ? 0 : -1;
}
