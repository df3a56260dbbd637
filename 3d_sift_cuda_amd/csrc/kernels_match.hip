/*
 * kernels_match.hip -- exact k-nearest-neighbour search over 64-component descriptors for gfx950 (MI355X): the search
 * step of the reference's matcher, SURVEY.md section 8f-3.
 *
 * Reference: featMatchMultiple indexes the descriptors (m_pfPC, 64 floats holding the ranks 0..63 that
 * NormalizeDataRankedPCs left, R/src_common/MultiScale.cpp:207-233) of every feature of every input image in one FLANN
 * kd-tree forest (8 trees, 64 checks: R/feat_common/featMatchUtilities.cpp:1449-1455,1559) and asks it for the iNeighbors
 * nearest neighbours of every feature (:1612).  FLANN's search is approximate and randomised; FLANN itself is not in
 * /root/reference (fetched at build time by the reference's CMake).  This is the same search done exactly:
 *
 *     d2(q, b) = |q|^2 + |b|^2 - 2 q.b      over int8 components
 *
 * with the Gram matrix q.b on the matrix cores (v_mfma_i32_32x32x32_i8: integer arithmetic, so "exact" is literal) and a
 * per-lane running top-k.  Unlike every other kernel of this library this one IS matrix-shaped: 2 * 64 * n_q * n_db
 * integer operations on 64 (n_q + n_db) bytes.
 *
 * Mapping.  A workgroup of four wavefronts owns 128 queries (32 per wavefront: the B operand, loaded once, 2 x 16 bytes
 * per lane) and walks the database in tiles of 256 vectors staged through LDS (16 KB, double-buffered; the next tile is in
 * registers while this one is multiplied).  Per 32-vector subtile a wavefront issues two MFMAs (K = 2 x 32) with the
 * database vectors as rows: the 32 x 32 result has the QUERY on the lane (column = lane & 31) and sixteen database rows in
 * the lane's sixteen accumulator registers -- so a lane keeps the running top-k of ITS query in registers and never talks
 * to another lane: sixteen candidates per subtile, each compared with the lane's current k-th distance (an insertion is
 * rare after the first few tiles).  The two lanes that share a query (lane and lane + 32 see different rows) write their
 * lists to a scratch array; knn_merge_kernel merges them.  Order: ascending (distance, database index) -- ties to the
 * lower index -- which makes the result unique.
 */
#include "sift3d_internal.h"

typedef int m_v4i __attribute__((ext_vector_type(4)));
typedef int m_v16i __attribute__((ext_vector_type(16)));

#define KNN_DIM 64
#define KNN_TILE 256    /* database vectors per LDS tile: one per thread of the workgroup */
#define KNN_QW 32       /* queries per wavefront */
#define KNN_QWG 128     /* queries per workgroup */
#define KNN_BIG 0x7fffffff

/* |v|^2 per vector (one thread per vector) */
__global__ void knn_norms_kernel(const signed char *__restrict__ v, long long n, int *__restrict__ norms)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const m_v4i *p = reinterpret_cast<const m_v4i *>(v + i * KNN_DIM);
    int s = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const m_v4i x = p[w];
        const int xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = (int)(signed char)((xs[e] >> (8 * b)) & 0xff);
                s += c * c;
            }
    }
    norms[i] = s;
}

/* KK: list length kept per lane (>= k).  part: [query][2][KK] pairs (dist2, index).
 * CN: every database vector has the same squared norm (rank descriptors: always 85 344) -- then the distance is a decreasing
 * function of q.b alone and the common case needs no arithmetic on the candidates at all: the largest of a lane's sixteen
 * dot products (a tree of eight three-operand maxima) against one per-lane threshold. */
template <int KK, bool CN>
__global__ __launch_bounds__(256) void knn_search_kernel(const signed char *__restrict__ db, const int *__restrict__ db_norm, long long n_db,
                                                         const signed char *__restrict__ q, const int *__restrict__ q_norm, long long n_q,
                                                         int const_norm, int *__restrict__ part_d, int *__restrict__ part_i)
{
    __shared__ __attribute__((aligned(16))) signed char tile[2][KNN_TILE * KNN_DIM];
    __shared__ __attribute__((aligned(16))) int tnorm[2][KNN_TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long long qi = (long long)blockIdx.x * KNN_QWG + wave * KNN_QW + r; /* this lane's query */
    const bool qok = qi < n_q;
    /* B operand: the query's bytes [32 s + 16 h, +16) for k-step s */
    m_v4i bq[2];
#pragma unroll
    for (int s = 0; s < 2; s++) bq[s] = qok ? *reinterpret_cast<const m_v4i *>(q + qi * KNN_DIM + 32 * s + 16 * h) : m_v4i(0);
    const int qn = qok ? q_norm[qi] : 0;
    int bd[KK], bi[KK];
#pragma unroll
    for (int j = 0; j < KK; j++) {
        bd[j] = KNN_BIG;
        bi[j] = -1;
    }
    /* a candidate enters the list iff  2 q.b - |b|^2 > |q|^2 - (k-th distance)   (CN: iff q.b > gate, the same halved) */
    int gate = -KNN_BIG;
    /* staging: thread t carries 64 bytes of the tile (vector t) and one norm */
    const long long ntiles = (n_db + KNN_TILE - 1) / KNN_TILE;
    m_v4i st[4];
    int stn = 0;
    auto fetch = [&](long long t) {
        const long long v = t * KNN_TILE + tid;
        const bool ok = v < n_db;
        const m_v4i *p = reinterpret_cast<const m_v4i *>(db + v * KNN_DIM);
#pragma unroll
        for (int w = 0; w < 4; w++) st[w] = ok ? p[w] : m_v4i(0);
        if (!CN) stn = ok ? -db_norm[v] : -(KNN_BIG / 2); /* negated; a vector past the end can never be among the nearest */
    };
    auto stash = [&](int buf) {
        m_v4i *p = reinterpret_cast<m_v4i *>(&tile[buf][tid * KNN_DIM]);
#pragma unroll
        for (int w = 0; w < 4; w++) p[w] = st[w];
        if (!CN) tnorm[buf][tid] = stn;
    };
    auto gram = [&](int buf, int sub) -> m_v16i { /* A operand: database vector (row) sub * 32 + r, bytes [32 s + 16 h, +16) */
        const signed char *row = &tile[buf][(sub * 32 + r) * KNN_DIM + 16 * h];
        const m_v4i a0 = *reinterpret_cast<const m_v4i *>(row), a1 = *reinterpret_cast<const m_v4i *>(row + 32);
        m_v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[0], acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[1], acc, 0, 0, 0);
    };
    /* accumulator register g holds row (g & 3) + 8 (g >> 2) + 4 h of the subtile */
    auto take = [&](const m_v16i &acc, int buf, int sub, long long t) {
        int sv[16];
        bool any;
        if constexpr (CN) {
#pragma unroll
            for (int g = 0; g < 16; g++) sv[g] = acc[g];
            int m[6];
#pragma unroll
            for (int u = 0; u < 5; u++) m[u] = max(max(sv[3 * u], sv[3 * u + 1]), sv[3 * u + 2]);
            m[5] = sv[15];
            any = max(max(max(m[0], m[1]), m[2]), max(max(m[3], m[4]), m[5])) > gate;
        } else {
            any = false;
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const m_v4i nn = *reinterpret_cast<const m_v4i *>(&tnorm[buf][sub * 32 + 8 * g4 + 4 * h]); /* -|b|^2, four per read */
                const int nv[4] = {nn.x, nn.y, nn.z, nn.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    sv[4 * g4 + e] = (acc[4 * g4 + e] << 1) + nv[e];
                    any = any || sv[4 * g4 + e] > gate;
                }
            }
        }
        if (__ballot(any)) { /* only a wavefront with a hit somewhere enters the insertion code */
            const long long base = t * KNN_TILE + sub * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int d = CN ? qn + const_norm - 2 * sv[g] : qn - sv[g];
                const long long idx = base + (g & 3) + 8 * (g >> 2);
                if (d < bd[KK - 1] && idx < n_db) { /* insert behind every entry with a distance <= d (indices arrive ascending) */
                    int cd = d, ci = (int)idx;
                    bool shifting = false; /* once the new entry is in, everything behind it moves down one place (a displaced
                                            * entry that ties with its successor must stay in front of it) */
#pragma unroll
                    for (int j = 0; j < KK; j++) {
                        const bool sw = shifting || cd < bd[j];
                        shifting = sw;
                        const int td = bd[j], ti = bi[j];
                        bd[j] = sw ? cd : td;
                        bi[j] = sw ? ci : ti;
                        cd = sw ? td : cd;
                        ci = sw ? ti : ci;
                    }
                }
            }
            if (bd[KK - 1] == KNN_BIG) gate = -KNN_BIG;
            else gate = CN ? (qn + const_norm - bd[KK - 1]) >> 1 : qn - bd[KK - 1]; /* CN: 2 q.b > x  <=>  q.b > floor(x / 2) */
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (long long t = 0; t < ntiles; t++) {
        const int buf = (int)(t & 1);
        if (t + 1 < ntiles) fetch(t + 1); /* in flight while this tile is multiplied */
        /* the matrix cores work on subtile s + 1 while the vector unit looks at the results of subtile s */
        m_v16i acc0 = gram(buf, 0), acc1;
#pragma unroll
        for (int sub = 0; sub < KNN_TILE / 32; sub += 2) {
            acc1 = gram(buf, sub + 1);
            take(acc0, buf, sub, t);
            if (sub + 2 < KNN_TILE / 32) acc0 = gram(buf, sub + 2);
            take(acc1, buf, sub + 1, t);
        }
        if (t + 1 < ntiles) stash(buf ^ 1); /* the other buffer was last read one iteration ago: every wavefront has passed the barrier below since */
        __syncthreads();
    }
    if (qok) {
        int *pd = part_d + (qi * 2 + h) * KK, *pi = part_i + (qi * 2 + h) * KK;
#pragma unroll
        for (int j = 0; j < KK; j++) {
            pd[j] = bd[j];
            pi[j] = bi[j];
        }
    }
}

/* two ascending lists per query -> the k best of their union, ascending by (distance, index) */
__global__ void knn_merge_kernel(const int *__restrict__ part_d, const int *__restrict__ part_i, long long n_q, int KK, int k,
                                 int *__restrict__ out_i, int *__restrict__ out_d)
{
    const long long qi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= n_q) return;
    const int *d0 = part_d + qi * 2 * KK, *d1 = d0 + KK, *i0 = part_i + qi * 2 * KK, *i1 = i0 + KK;
    int a = 0, b = 0;
    for (int j = 0; j < k; j++) {
        const bool ta = a < KK && i0[a] >= 0, tb = b < KK && i1[b] >= 0;
        bool pick_a;
        if (ta && tb) pick_a = d0[a] < d1[b] || (d0[a] == d1[b] && i0[a] < i1[b]);
        else pick_a = ta;
        if (!ta && !tb) {
            out_i[qi * k + j] = -1;
            out_d[qi * k + j] = KNN_BIG;
            continue;
        }
        if (pick_a) {
            out_i[qi * k + j] = i0[a];
            out_d[qi * k + j] = d0[a];
            a++;
        } else {
            out_i[qi * k + j] = i1[b];
            out_d[qi * k + j] = d1[b];
            b++;
        }
    }
}

hipError_t sift3d_launch_knn_norms(hipStream_t s, const signed char *v, int64_t n, int *norms)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, (long long)n, norms);
    return hipGetLastError();
}

int sift3d_knn_list_length(int k) { return k <= 8 ? 8 : (k <= 16 ? 16 : (k <= 32 ? 32 : 0)); }

/* const_norm >= 0: every database vector has this squared norm (the caller checked) */
hipError_t sift3d_launch_knn(hipStream_t s, const signed char *db, const int *db_norm, int64_t n_db, const signed char *q, const int *q_norm,
                             int64_t n_q, int k, int const_norm, int *part_d, int *part_i, int *out_i, int *out_d)
{
    if (n_q <= 0) return hipSuccess;
    const int KK = sift3d_knn_list_length(k);
    if (KK == 0 || k < 1) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n_q + KNN_QWG - 1) / KNN_QWG));
#define KNN_LAUNCH(KK_, CN_)                                                                                                       \
    hipLaunchKernelGGL((knn_search_kernel<KK_, CN_>), grid, dim3(256), 0, s, db, db_norm, (long long)n_db, q, q_norm, (long long)n_q, \
                       const_norm, part_d, part_i)
    if (const_norm >= 0) {
        if (KK == 8) KNN_LAUNCH(8, true);
        else if (KK == 16) KNN_LAUNCH(16, true);
        else KNN_LAUNCH(32, true);
    } else {
        if (KK == 8) KNN_LAUNCH(8, false);
        else if (KK == 16) KNN_LAUNCH(16, false);
        else KNN_LAUNCH(32, false);
    }
#undef KNN_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((n_q + 255) / 256)), dim3(256), 0, s, part_d, part_i, (long long)n_q, KK, k, out_i, out_d);
    return hipGetLastError();
}
