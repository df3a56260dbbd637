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
 * Mapping.  A workgroup of four wavefronts owns 128 G queries (G = 1 in the product): a wavefront keeps G groups of 32 as B operands in registers
 * (loaded once, 2 x 16 bytes per lane and group) and every workgroup walks its segment of the database in tiles of 256
 * vectors staged through LDS (16 KB, double-buffered; the next tile is in registers while this one is multiplied).  Per
 * 32-vector subtile a wavefront reads the A operand once (two ds_read_b128) and issues two MFMAs per group (K = 2 x 32) with
 * the database vectors as rows: the 32 x 32 result has the QUERY on the lane (column = lane & 31) and sixteen database rows
 * in the lane's sixteen accumulator registers -- so a lane keeps the running top-k of ITS queries in registers and never
 * talks to another lane: sixteen candidates per subtile and group, their maximum compared with one per-lane threshold (an
 * insertion is rare once the lists have settled).  The two lanes that share a query (lane and lane + 32 see different
 * rows), and the workgroups that share it (one per database segment), write their lists to a scratch array;
 * knn_merge_kernel merges them.  Order: ascending (distance, database index) -- ties to the lower index -- which makes the
 * result unique.
 *
 * The lists.  An entry is ONE double: distance * 2^31 + index (distances stay below 2^22 -- 64 components of at most 127 --
 * and indices below 2^31, so every key is an integer below 2^53 and exact).  Keys are unique and their order is
 * (distance, index), so a sorted insertion is a chain of v_min_f64 / v_max_f64 pairs, two instructions per place, with no
 * case for ties and no dependence on the order of arrival.  That last property pays for a two-entry QUEUE per lane: a
 * candidate that beats the lane's threshold is only noted (its dot product and row), and the lists are brought up to date --
 * queue entries inserted, thresholds recomputed from both lanes of a pair -- when some lane of the wavefront has filled
 * its queue, about every tenth candidate; a lane whose queue is full inserts directly.  The count of candidates that beat
 * a running k-th distance is k ln(N / k) per query whatever one does (records of a random sequence: some sixty per query
 * here), so the vector instructions spent on each of them, not their number, is what the insertion path can save.
 *
 * The LDS image.  A row is 64 bytes = four 16-byte chunks, and the sixteen lanes one ds_read_b128 cycle serves read the
 * same chunk of sixteen different rows: in a linear image they fall on four of the sixteen 16-byte slots of the 256-byte
 * bank row (slot = 4 row + chunk mod 16), a 4-way conflict on every read -- with four wavefronts per SIMD that alone capped
 * the first form of this kernel at half the matrix rate.  Chunk c of row R therefore lives at chunk position
 * c ^ ((R >> 2) & 3): rows equal mod 4 differ in that term within every lane group of the instruction (MI355X_MICROARCH.md,
 * LDS: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...), so the sixteen lanes cover the sixteen slots.  The tile is staged
 * with thread t carrying chunk t & 3 of rows (t >> 2) + 64 j: a wavefront's load covers 1 KB of contiguous memory and its
 * eight-lane store groups cover eight slots.
 */
#include <type_traits>
#include <utility>

#include "sift3d_internal.h"

typedef int m_v4i __attribute__((ext_vector_type(4)));
typedef int m_v16i __attribute__((ext_vector_type(16)));

#define KNN_DIM 64
#define KNN_TILE 256    /* database vectors per LDS tile */
#define KNN_QW 32       /* queries per wavefront and group */
#define KNN_BIG 0x7fffffff
#define KNN_EMPTY 4611686018427387904.0 /* 2^62: above every key */
#define KNN_PAD_NORM (1 << 21)          /* "squared norm" of the rows past the end of the database: above every real one, and
                                         * their distances still below 2^22 */
#define KNN_MAX_SEGMENTS 8
/* wavefronts per SIMD the register allocation is asked to allow (the matrix instructions take their VGPR form at two or more) */
#ifndef KNN_WAVES
#define KNN_WAVES(KK, CN, G) 2
#endif

/* |v|^2 per vector (one thread per vector), and what the host used to find by reading every byte itself: stats[0] = the
 * lowest flat index of a component outside 0..127 (the matrix cores take signed bytes), stats[1] / stats[2] = the smallest
 * and the largest squared norm (equal: the constant-norm kernel applies).  The caller presets {~0, ~0, 0}. */
__global__ void knn_norms_kernel(const signed char *__restrict__ v, long long n, int *__restrict__ norms, unsigned long long *__restrict__ stats)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int s = 0;
    unsigned long long bad = ~0ull;
    if (i < n) {
        const m_v4i *p = reinterpret_cast<const m_v4i *>(v + i * KNN_DIM);
#pragma unroll
        for (int w = 3; w >= 0; w--) {
            const m_v4i x = p[w];
            const int xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 3; e >= 0; e--)
#pragma unroll
                for (int b = 3; b >= 0; b--) {
                    const int c = (int)(signed char)((xs[e] >> (8 * b)) & 0xff);
                    s += c * c;
                    if (c < 0) bad = (unsigned long long)(i * KNN_DIM + 16 * w + 4 * e + b); /* descending: the lowest index stays */
                }
        }
        norms[i] = s;
    }
    unsigned long long lo = i < n ? (unsigned long long)s : ~0ull, hi = i < n ? (unsigned long long)s : 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { /* one set of atomics per wavefront */
        const unsigned long long ob = __shfl_xor(bad, d), ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
        bad = ob < bad ? ob : bad;
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad != ~0ull) atomicMin(stats, bad);
        atomicMin(stats + 1, lo);
        atomicMax(stats + 2, hi);
    }
}

/* f(integral_constant<int, k>) for a wave-uniform 1 <= k <= KK (k above KK: KK) */
template <class F, int... Ks>
__device__ __forceinline__ void knn_dispatch_rank_seq(int k, F &&f, std::integer_sequence<int, Ks...>)
{
    constexpr int KK = (int)sizeof...(Ks);
    ((void)((k == Ks + 1 || (Ks + 1 == KK && k > KK)) ? (f(std::integral_constant<int, Ks + 1>{}), 0) : 0), ...);
}
template <int KK, class F>
__device__ __forceinline__ void knn_dispatch_rank(int k, F &&f)
{
    knn_dispatch_rank_seq(k, f, std::make_integer_sequence<int, KK>{});
}

/* KK: list length kept per lane and query (>= k).  G: groups of 32 queries per wavefront.
 * part: [query][segment][2][KK] pairs (dist2, index); blockIdx.y = database segment (tiles [seg * tps, (seg + 1) * tps)).
 * CN: every database vector has the same squared norm (rank descriptors: always 85 344) -- then the distance is a decreasing
 * function of q.b alone and the common case needs no arithmetic on the candidates at all: the largest of a lane's sixteen
 * dot products (a tree of eight three-operand maxima) against one per-lane threshold. */
/* AHEAD: subtiles the matrix cores run in front of the vector unit.  1 (the product): the products of subtile s + 1 are issued,
 * then the results of s are looked at.  2 (development builds, round-4 review item 7): three accumulator sets, the products
 * of s + 2 issued before s is looked at -- measured, DESIGN.md section 7a. */
template <int KK, bool CN, int G, int AHEAD = 1>
__global__ __launch_bounds__(256, KNN_WAVES(KK, CN, G)) void knn_search_kernel(const signed char *__restrict__ db, const int *__restrict__ db_norm, long long n_db,
                                                         const signed char *__restrict__ q, const int *__restrict__ q_norm, long long n_q,
                                                         int const_norm, int k, long long tiles_per_segment, int *__restrict__ part_d,
                                                         int *__restrict__ part_i)
{
    __shared__ __attribute__((aligned(256))) signed char tile[2][KNN_TILE * KNN_DIM];
    __shared__ __attribute__((aligned(16))) int tnorm[2][CN ? 4 : KNN_TILE]; /* CN: unused */
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nseg = (int)gridDim.y, seg = (int)blockIdx.y;
    long long qi[G]; /* this lane's queries */
    bool qok[G];
    m_v4i bq[G][2];  /* B operand: the query's bytes [32 s + 16 h, +16) for k-step s */
    int qn[G], gate[G];
    double lst[G][KK];    /* ascending keys; KNN_EMPTY: no entry */
    int qs[G][2], qr[G][2], qc[G]; /* the queue: dot product (CN) or 2 q.b - |b|^2, row index less 4 h; entries in use */
#pragma unroll
    for (int g = 0; g < G; g++) {
        qi[g] = ((long long)blockIdx.x * 4 + wave) * (KNN_QW * G) + g * KNN_QW + r;
        qok[g] = qi[g] < n_q;
#pragma unroll
        for (int s = 0; s < 2; s++) bq[g][s] = qok[g] ? *reinterpret_cast<const m_v4i *>(q + qi[g] * KNN_DIM + 32 * s + 16 * h) : m_v4i(0);
        qn[g] = qok[g] ? q_norm[qi[g]] : 0;
#pragma unroll
        for (int j = 0; j < KK; j++) lst[g][j] = KNN_EMPTY;
        qc[g] = 0;
        qs[g][0] = qs[g][1] = qr[g][0] = qr[g][1] = 0;
        /* a candidate enters the list iff  2 q.b - |b|^2 > |q|^2 - (k-th distance)   (CN: iff q.b > gate, the same halved) */
        gate[g] = -KNN_BIG;
    }
    const long long ntiles = (n_db + KNN_TILE - 1) / KNN_TILE;
    const long long t_first = seg * tiles_per_segment, t_end = min(ntiles, t_first + tiles_per_segment);
    /* staging: thread t carries chunk t & 3 of rows (t >> 2) + 64 j of the tile, and the norm of row t */
    const int sc = tid & 3, srow = tid >> 2;
    m_v4i st[4];
    int stn = 0;
    auto fetch = [&](long long t) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long long v = t * KNN_TILE + srow + 64 * j;
            st[j] = v < n_db ? *reinterpret_cast<const m_v4i *>(db + v * KNN_DIM + 16 * sc) : m_v4i(0);
        }
        if (!CN) {
            const long long v = t * KNN_TILE + tid;
            stn = v < n_db ? -db_norm[v] : -KNN_PAD_NORM; /* negated; a row past the end is farther than every real one */
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = srow + 64 * j;
            *reinterpret_cast<m_v4i *>(&tile[buf][row * KNN_DIM + 16 * (sc ^ ((row >> 2) & 3))]) = st[j];
        }
        if (!CN) tnorm[buf][tid] = stn;
    };
    /* A operand: database vector (row) sub * 32 + r, bytes [32 s + 16 h, +16) = chunk 2 s + h, at its swizzled position */
    const int sw = (r >> 2) & 3;
    const int a_off0 = r * KNN_DIM + 16 * (h ^ sw), a_off1 = r * KNN_DIM + 16 * ((2 + h) ^ sw);
    auto dist_of = [&](double key) __attribute__((always_inline)) -> int { return key >= KNN_EMPTY ? KNN_BIG : (int)(key * (1.0 / 2147483648.0)); };
    auto insert = [&](int g, int sval, int row) __attribute__((always_inline)) {
        const int d = CN ? qn[g] + const_norm - 2 * sval : qn[g] - sval;
        double key = (double)d * 2147483648.0 + (double)row;
#pragma unroll
        for (int j = 0; j < KK; j++) {
            const double lo = __builtin_fmin(lst[g][j], key);
            key = __builtin_fmax(lst[g][j], key);
            lst[g][j] = lo;
        }
    };
    /* The threshold behind a list, from the k-th distance (k as asked for, not the list's length) of this lane's list AND
     * its partner's (lane ^ 32 holds the same query's list over the other rows): with k entries at or below U already known
     * between the two, anything above U is out of the final top k, whichever lane it would have entered.  Any
     * max(A[i-1], B[k-i-1]) bounds the k-th of the union of two ascending lists A and B; three of them are used: i = k (this
     * lane's k-th), i = 0 (the partner's), i = k / 2 (the middle: about where the k-th of two halves of one candidate
     * stream lies).  A tie with an entry of the partner may still win on the index, so the bound from the partner is
     * "<= U": candidates enter iff d < min(own k-th, U + 1).  All lanes run this together, so each sees the partner's list
     * as it stands -- which only improves afterwards. */
    auto new_gate = [&](int g, auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value, KH = K / 2;
        const int ak = dist_of(lst[g][K - 1]);
        int ucross = __shfl_xor(ak, 32);
        if constexpr (KH >= 1) ucross = min(ucross, max(dist_of(lst[g][KH - 1]), __shfl_xor(dist_of(lst[g][K - KH - 1]), 32)));
        int limit = ak;
        if (ucross != KNN_BIG && ucross + 1 < limit) limit = ucross + 1;
        if (limit != KNN_BIG) gate[g] = max(gate[g], CN ? (qn[g] + const_norm - limit) >> 1 : qn[g] - limit); /* CN: 2 q.b > x  <=>  q.b > floor(x / 2) */
    };
    /* queue -> lists, then the thresholds (every lane of the wavefront) */
    auto settle = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (j < qc[g]) insert(g, qs[g][j], qr[g][j]);
        qc[g] = 0;
        knn_dispatch_rank<KK>(k, [&](auto kc) __attribute__((always_inline)) { new_gate(g, kc); });
    };
    /* accumulator register e holds row (e & 3) + 8 (e >> 2) + 4 h of the subtile */
    auto take = [&](const m_v16i &acc, int g, int buf, int sub, long long t) __attribute__((always_inline)) {
        constexpr int NG = CN ? 6 : 4, GW = CN ? 3 : 4; /* candidates in groups of GW: one maximum each, then one over the groups */
        int sv[16], m[NG];
        if constexpr (CN) {
#pragma unroll
            for (int e = 0; e < 16; e++) sv[e] = acc[e];
#pragma unroll
            for (int u = 0; u < 5; u++) m[u] = max(max(sv[3 * u], sv[3 * u + 1]), sv[3 * u + 2]);
            m[5] = sv[15];
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const m_v4i nn = *reinterpret_cast<const m_v4i *>(&tnorm[buf][sub * 32 + 8 * u + 4 * h]); /* -|b|^2, four per read */
                const int nv[4] = {nn.x, nn.y, nn.z, nn.w};
#pragma unroll
                for (int e = 0; e < 4; e++) sv[4 * u + e] = (acc[4 * u + e] << 1) + nv[e];
                m[u] = max(max(sv[4 * u], sv[4 * u + 1]), max(sv[4 * u + 2], sv[4 * u + 3]));
            }
        }
        int mall = max(max(m[0], m[1]), max(m[2], m[3]));
        if constexpr (NG == 6) mall = max(max(mall, m[4]), m[5]);
        if (__builtin_expect(__ballot(mall > gate[g]) != 0, 0)) { /* only a wavefront with a candidate somewhere leaves the common path (which
                                                                   * falls through: a taken branch costs the wavefront an instruction fetch) */
            /* the row's index less 4 h: the same for every lane, so it costs the common path nothing (kept in scalar
             * registers); the lane's 4 h is added when the lists are written */
            const int base = __builtin_amdgcn_readfirstlane((int)(t * KNN_TILE) + sub * 32);
#pragma unroll
            for (int u = 0; u < NG; u++) {
                if (!__ballot(m[u] > gate[g])) continue; /* nothing in this group, for any lane */
#pragma unroll
                for (int e = GW * u; e < GW * u + GW && e < 16; e++) {
                    /* No test for rows past the end of the database here (the compiler hoists it into the common path: sixteen
                     * index computations and compares per subtile).  Such rows are zero vectors farther from a query than
                     * every real row: they only ever take places no real row wants, at the tail of a list, and
                     * knn_merge_kernel ends a list at the first. */
                    if (sv[e] > gate[g]) {
                        const int row = base + (e & 3) + 8 * (e >> 2);
                        if (qc[g] < 2) { /* note it */
                            qs[g][1] = qs[g][0];
                            qr[g][1] = qr[g][0];
                            qs[g][0] = sv[e];
                            qr[g][0] = row;
                            qc[g]++;
                        } else insert(g, sv[e], row); /* the queue is full: straight into the list */
                    }
                }
            }
            if (__ballot(qc[g] >= 2)) settle(g);
        }
    };
    auto gram = [&](const m_v4i &a0, const m_v4i &a1, int g) __attribute__((always_inline)) -> m_v16i {
        m_v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[g][0], acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[g][1], acc, 0, 0, 0);
    };
    if (t_first < t_end) {
        fetch(t_first);
        stash(0);
    }
    __syncthreads();
    for (long long t = t_first; t < t_end; t++) {
        const int buf = (int)((t - t_first) & 1);
        if (t + 1 < t_end) fetch(t + 1); /* in flight while this tile is multiplied */
        const signed char *tb = &tile[buf][0];
        /* One subtile ahead: the matrix cores work on subtile s + 1 while the vector unit looks at the results of s.  Two
         * subtiles per trip, so that the two accumulator sets keep their registers.  (Asking for the A operand another
         * subtile earlier costs 14 - 60 registers and gains nothing: four wavefronts per SIMD cover the LDS latency.) */
        /* the two operand addresses march with the loop (one add each per trip); the subtile inside a trip is an immediate */
        const signed char *p0 = tb + a_off0, *p1 = tb + a_off1;
        if constexpr (AHEAD == 2) {
            constexpr int NS = KNN_TILE / 32;
            m_v16i acc3[3][G];
            auto issue = [&](int sub, int slot) __attribute__((always_inline)) {
                const m_v4i x0 = *reinterpret_cast<const m_v4i *>(p0 + sub * 32 * KNN_DIM), x1 = *reinterpret_cast<const m_v4i *>(p1 + sub * 32 * KNN_DIM);
#pragma unroll
                for (int g = 0; g < G; g++) acc3[slot][g] = gram(x0, x1, g);
            };
            issue(0, 0);
            issue(1, 1);
#pragma unroll
            for (int sub = 0; sub < NS; sub++) {
                if (sub + 2 < NS) issue(sub + 2, (sub + 2) % 3);
#pragma unroll
                for (int g = 0; g < G; g++) take(acc3[sub % 3][g], g, buf, sub, t);
            }
            if (t + 1 < t_end) stash(buf ^ 1);
            __syncthreads();
            continue;
        }
        m_v16i acc_a[G], acc_b[G];
        {
            const m_v4i x0 = *reinterpret_cast<const m_v4i *>(p0), x1 = *reinterpret_cast<const m_v4i *>(p1);
#pragma unroll
            for (int g = 0; g < G; g++) acc_a[g] = gram(x0, x1, g);
        }
#pragma unroll 1
        for (int sub = 0; sub < KNN_TILE / 32; sub += 2) {
            {
                const m_v4i x0 = *reinterpret_cast<const m_v4i *>(p0 + 32 * KNN_DIM), x1 = *reinterpret_cast<const m_v4i *>(p1 + 32 * KNN_DIM);
#pragma unroll
                for (int g = 0; g < G; g++) acc_b[g] = gram(x0, x1, g);
            }
#pragma unroll
            for (int g = 0; g < G; g++) take(acc_a[g], g, buf, sub, t);
            if (sub + 2 < KNN_TILE / 32) {
                const m_v4i x0 = *reinterpret_cast<const m_v4i *>(p0 + 64 * KNN_DIM), x1 = *reinterpret_cast<const m_v4i *>(p1 + 64 * KNN_DIM);
#pragma unroll
                for (int g = 0; g < G; g++) acc_a[g] = gram(x0, x1, g);
            }
#pragma unroll
            for (int g = 0; g < G; g++) take(acc_b[g], g, buf, sub + 1, t);
            p0 += 64 * KNN_DIM;
            p1 += 64 * KNN_DIM;
#ifdef KNN_EXTRA_BARRIER /* experiment: what a workgroup barrier costs here */
            if (sub == 2) __syncthreads();
#endif
        }
        if (t + 1 < t_end) stash(buf ^ 1); /* the other buffer was last read one iteration ago: every wavefront has passed the barrier below since */
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        settle(g);
        if (qok[g]) {
            int *pd = part_d + ((qi[g] * nseg + seg) * 2 + h) * KK, *pi = part_i + ((qi[g] * nseg + seg) * 2 + h) * KK;
#pragma unroll
            for (int j = 0; j < KK; j++) {
                const double key = lst[g][j];
                const int d = dist_of(key);
                pd[j] = d;
                pi[j] = key >= KNN_EMPTY ? -1 : (int)(key - (double)d * 2147483648.0) + 4 * h;
            }
        }
    }
}

/* nlists ascending lists per query -> the k best of their union, ascending by (distance, index) */
__global__ void knn_merge_kernel(const int *__restrict__ part_d, const int *__restrict__ part_i, long long n_q, int n_db, int KK, int nlists, int k,
                                 int *__restrict__ out_i, int *__restrict__ out_d)
{
    const long long qi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= n_q) return;
    const int *pd = part_d + qi * nlists * KK, *pi = part_i + qi * nlists * KK;
    int pos[2 * KNN_MAX_SEGMENTS];
    for (int l = 0; l < nlists; l++) pos[l] = 0;
    for (int j = 0; j < k; j++) {
        int best = -1, bdist = KNN_BIG, bidx = -1;
        for (int l = 0; l < nlists; l++) {
            if (pos[l] >= KK) continue;
            const int d = pd[l * KK + pos[l]], i = pi[l * KK + pos[l]];
            if (i < 0 || i >= n_db) continue; /* the list has ended (rows past the end of the database: see the search kernel) */
            if (best < 0 || d < bdist || (d == bdist && i < bidx)) {
                best = l;
                bdist = d;
                bidx = i;
            }
        }
        out_i[qi * k + j] = bidx;
        out_d[qi * k + j] = bdist;
        if (best >= 0) pos[best]++;
    }
}
/* stats: three words the caller has preset to {~0, ~0, 0} (knn_norms_kernel) */
hipError_t sift3d_launch_knn_norms(hipStream_t s, const signed char *v, int64_t n, int *norms, unsigned long long *stats)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, (long long)n, norms, stats);
    return hipGetLastError();
}

int sift3d_knn_list_length(int k) { return k <= 8 ? 8 : (k <= 16 ? 16 : (k <= 32 ? 32 : 0)); }

static int g_knn_dev_groups = 0, g_knn_dev_segments = 0; /* -DSIFT3D_DEV builds: sift3d_dev_knn_plan */
#ifdef SIFT3D_DEV
static int g_knn_dev_ahead = 1;
extern "C" void sift3d_dev_knn_plan(int groups, int segments)
{
    g_knn_dev_groups = groups;
    g_knn_dev_segments = segments;
}
extern "C" void sift3d_dev_knn_ahead(int ahead) { g_knn_dev_ahead = ahead == 2 ? 2 : 1; }
#endif

/* How a search is cut.  One group of 32 queries per wavefront (two were slower at every size tried: the insertion path, not
 * the LDS reads, is what a wavefront spends its time on, and half as many workgroups fill the chip worse).  Database
 * segments only where the queries alone give the chip few workgroups: every segment fills its lists from empty, which costs
 * insertions, so segments are added only up to about 600 workgroups and never below 16 tiles each (measured, k = 5:
 * 20 000 x 20 000 best with 4 segments, 0.167 against 0.231 ms with one; 50 000^2 with 2, 0.369 against 0.430;
 * 100 000^2 and 200 000^2 with one). */
void sift3d_knn_plan(int64_t n_db, int64_t n_q, int k, int *groups, int *segments)
{
    (void)k;
    const int64_t ntiles = (n_db + KNN_TILE - 1) / KNN_TILE, qblocks = (n_q + 127) / 128;
    int sg = 1;
    while (sg < KNN_MAX_SEGMENTS && qblocks * sg < 600 && ntiles / (sg + 1) >= 16) sg++;
    *groups = g_knn_dev_groups > 0 && g_knn_dev_groups <= 2 ? g_knn_dev_groups : 1; /* 2: development builds only, lists of 8 */
    *segments = g_knn_dev_segments > 0 && g_knn_dev_segments <= KNN_MAX_SEGMENTS ? g_knn_dev_segments : sg;
}

/* const_norm >= 0: every database vector has this squared norm (the caller checked).  part_d / part_i: n_q * 2 * segments *
 * list length ints each, with (groups, segments) from sift3d_knn_plan. */
hipError_t sift3d_launch_knn(hipStream_t s, const signed char *db, const int *db_norm, int64_t n_db, const signed char *q, const int *q_norm,
                             int64_t n_q, int k, int const_norm, int groups, int segments, int *part_d, int *part_i, int *out_i, int *out_d)
{
    if (n_q <= 0) return hipSuccess;
    const int KK = sift3d_knn_list_length(k);
    if (KK == 0 || k < 1 || segments < 1 || segments > KNN_MAX_SEGMENTS || groups < 1 || groups > 2) return hipErrorInvalidValue;
#ifndef SIFT3D_DEV
    if (groups != 1) return hipErrorInvalidValue;
#else
    if (groups == 2 && KK != 8) return hipErrorInvalidValue;
#endif
    const int64_t ntiles = (n_db + KNN_TILE - 1) / KNN_TILE;
    const long long tps = (long long)((ntiles + segments - 1) / segments);
    const dim3 grid((unsigned)((n_q + 128 * groups - 1) / (128 * groups)), (unsigned)segments);
#define KNN_LAUNCH(KK_, CN_, G_)                                                                                                       \
    hipLaunchKernelGGL((knn_search_kernel<KK_, CN_, G_>), grid, dim3(256), 0, s, db, db_norm, (long long)n_db, q, q_norm, (long long)n_q, \
                       const_norm, k, tps, part_d, part_i)
    if (const_norm >= 0) {
#ifdef SIFT3D_DEV /* two groups per wavefront: measured slower at every size; development builds keep it */
        if (KK == 8 && groups == 2) KNN_LAUNCH(8, true, 2);
        else if (KK == 8 && g_knn_dev_ahead == 2) /* the matrix cores two subtiles ahead (round-4 review item 7) */
            hipLaunchKernelGGL((knn_search_kernel<8, true, 1, 2>), grid, dim3(256), 0, s, db, db_norm, (long long)n_db, q, q_norm, (long long)n_q, const_norm, k,
                               tps, part_d, part_i);
        else
#endif
        if (KK == 8) KNN_LAUNCH(8, true, 1);
        else if (KK == 16) KNN_LAUNCH(16, true, 1);
        else KNN_LAUNCH(32, true, 1);
    } else {
#ifdef SIFT3D_DEV
        if (KK == 8 && groups == 2) KNN_LAUNCH(8, false, 2);
        else
#endif
        if (KK == 8) KNN_LAUNCH(8, false, 1);
        else if (KK == 16) KNN_LAUNCH(16, false, 1);
        else KNN_LAUNCH(32, false, 1);
    }
#undef KNN_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((n_q + 255) / 256)), dim3(256), 0, s, part_d, part_i, (long long)n_q, (int)n_db, KK, 2 * segments, k,
                       out_i, out_d);
    return hipGetLastError();
}
