/*
 * kernels_chain.hip -- the coarse octaves of the pyramid as ONE persistent launch (round 4).
 *
 * After the first two octaves the pyramid of a 512^3 volume is 2.4 million voxels in six octaves -- 1/56 of the first
 * octave's work -- built by some sixty dependent launches of 7 - 20 us each whose sum (0.7 ms alone, more beside the first
 * octave's extrema passes, whose grids every one of those launches queues behind) is what the extrema count waits for.
 * The remedy the round-3 review asked for is fewer launches, not another schedule: this kernel builds every level and DoG
 * level of those octaves (R/src_common/MultiScale.cpp:337-560, the octave loop), subsamples between them
 * (fioSubSampleInterpolate, R/src_common/FeatureIO.cpp:1474-1554) and runs their 26 + 27 + 27 extrema test
 * (MultiScale.cpp:1523-1569, 2260-2524, 1135-1318) with grid-wide barriers where a launch boundary used to be.  Being
 * resident from its first instruction it does not queue behind anyone, and its few workgroups (no LDS) share their compute
 * units with the extrema march's.
 *
 * Nothing here is tuned for bandwidth -- the data is L2 / Infinity-Cache resident and the time is barriers: per level an
 * x, a y and a z pass over 16-byte vectors with the taps in registers, every level stored (D_0 .. D_4: no lazy levels
 * here), one thread per voxel quartet.  Arithmetic contract as everywhere (GaussBlur3D.cpp:43-61): acc = 0, then
 * acc = acc + f[j] * v[j] for ascending j with separately rounded multiply and add, float32 between the passes, zeros
 * outside the volume (a tap that falls outside is skipped: adding +0 to a sum that started at +0 changes nothing).
 *
 * Octaves of at most SIFT3D_CHAIN_SOLO_VOX voxels need no grid: workgroup 0 builds them alone with workgroup barriers
 * while the others run the extrema test of the larger ones.
 */
#include "sift3d_internal.h"

typedef float c4f __attribute__((ext_vector_type(4)));
/* 1024 threads a workgroup: sixteen wavefronts of at most 128 registers leave every SIMD of their compute unit room for one
 * wavefront of the extrema march (234 registers) -- as a 256-thread workgroup would, which brings a quarter of the threads */
#define CHAIN_THREADS 512

#ifdef SIFT3D_DEV /* development build: workgroup 0 leaves the 100 MHz clock at every barrier (tools/chain_phases.py) */
__device__ unsigned long long g_chain_clk[256];
__device__ unsigned g_chain_nclk;
#define CHAIN_CLK()                                                                     \
    do {                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0 && g_chain_nclk < 256) g_chain_clk[g_chain_nclk++] = wall_clock64(); \
    } while (0)
extern "C" int sift3d_dev_chain_clocks(unsigned long long *out, int n)
{
    unsigned cnt = 0;
    if (hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_chain_nclk), sizeof cnt) != hipSuccess) return -1;
    if ((int)cnt > n) cnt = (unsigned)n;
    if (cnt && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_clk), sizeof(unsigned long long) * cnt) != hipSuccess) return -1;
    const unsigned zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_chain_nclk), &zero, sizeof zero);
    return (int)cnt;
}
#else
#define CHAIN_CLK() do { } while (0)
#endif

namespace {
struct chain_ctx {
    unsigned wg, G;     /* this workgroup and the number of workgroups that share the work of a phase */
    unsigned tid;
};

__device__ __forceinline__ c4f ld4(const float *p) { return *reinterpret_cast<const c4f *>(p); }
__device__ __forceinline__ void st4(float *p, c4f v) { *reinterpret_cast<c4f *>(p) = v; }

/* x pass: out(row, x) = sum_j f[j] * in(row, x + j - R); rows of pitch XP (a multiple of 4) whose columns [X, XP) hold
 * zeros in `in`.  The pad columns of `out` get whatever the filter gives there: only the y and z passes read them, column
 * by column, and the z pass puts zeros back. */
template <int R>
__device__ __forceinline__ void pass_x(const chain_ctx &k, const float *__restrict__ in, float *__restrict__ out, int XP, unsigned rows, const float *f)
{
    constexpr int H4 = ((R + 3) / 4) * 4, NV = 2 * (H4 / 4) + 1, U = 2 * R + 1;
    const unsigned xv = (unsigned)XP / 4u;
    const unsigned items = rows * xv;
    for (unsigned i = k.wg * CHAIN_THREADS + k.tid; i < items; i += k.G * CHAIN_THREADS) {
        const unsigned row = i / xv;
        const int x = (int)(i % xv) * 4;
        const float *p = in + (long long)row * XP;
        float w[NV * 4];
        c4f wv[NV];
        bool okx[NV];
#pragma unroll
        for (int q = 0; q < NV; q++) {
            const int gx = x - H4 + 4 * q;
            const bool ok = gx >= 0 && gx < XP;
            wv[q] = ld4(p + (ok ? gx : x));
            okx[q] = ok;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NV; q++) {
            const c4f v = okx[q] ? wv[q] : c4f(0.0f);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < U; j++) acc = acc + f[j] * w[H4 - R + e + j];
            o[e] = acc;
        }
        c4f r;
        r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
        st4(out + (long long)row * XP + x, r);
    }
}

/* y or z pass: out(i) = sum_j f[j] * in(i + (j - R) * stride) along an axis of length len; outer = the other slow axis.
 * LAST (the z pass of a level): columns [X, XP) are stored as zeros, the level goes to `lvl` when that is not NULL, and
 * dog = prev - level. */
template <int R, bool LAST>
__device__ __forceinline__ void pass_col(const chain_ctx &k, const float *__restrict__ in, float *__restrict__ lvl, const float *__restrict__ prev,
                                         float *__restrict__ dog, int XP, int X, int Y, int Z, bool along_z, const float *f)
{
    constexpr int U = 2 * R + 1;
    constexpr int B = 9;  /* taps per batch */
    constexpr int NI = 2; /* voxel quartets per round: every load of a round (NI x B) is issued before the first is used -- a
                           * phase is a handful of rounds per thread, each one trip to the L2 / Infinity Cache and back */
    const unsigned xv = (unsigned)XP / 4u;
    const long long XY = (long long)XP * Y;
    const unsigned items = xv * (unsigned)Y * (unsigned)Z; /* at most 2^19 quartets: 32-bit index arithmetic */
    const int len = along_z ? Z : Y;
    const long long stride = along_z ? XY : XP;
    const unsigned step = k.G * CHAIN_THREADS;
    for (unsigned i0 = k.wg * CHAIN_THREADS + k.tid; i0 < items; i0 += NI * step) {
        int x[NI], c[NI];
        long long at[NI];
        bool live[NI];
        c4f acc[NI];
#pragma unroll
        for (int n = 0; n < NI; n++) {
            const unsigned i = i0 + (unsigned)n * step;
            live[n] = i < items;
            const unsigned ii = live[n] ? i : i0;
            x[n] = (int)(ii % xv) * 4;
            const unsigned yz = ii / xv;
            const int y = (int)(yz % (unsigned)Y), z = (int)(yz / (unsigned)Y);
            c[n] = along_z ? z : y;
            at[n] = (long long)z * XY + (long long)y * XP + x[n];
            acc[n] = c4f(0.0f);
        }
#pragma unroll
        for (int j0 = 0; j0 < U; j0 += B) {
            c4f v[NI][B];
            bool okm[NI][B];
#pragma unroll
            for (int n = 0; n < NI; n++)
#pragma unroll
                for (int q = 0; q < B; q++) {
                    const int j = j0 + q;
                    if (j < U) {
                        const int cc = c[n] + j - R;
                        const bool ok = cc >= 0 && cc < len;
                        v[n][q] = ld4(in + at[n] + (ok ? (long long)(j - R) * stride : 0ll)); /* outside the volume: any valid address, then zero */
                        okm[n][q] = ok;
                    }
                }
            __builtin_amdgcn_sched_barrier(0); /* every load of the round is in flight before the first is waited for */
#pragma unroll
            for (int n = 0; n < NI; n++)
#pragma unroll
                for (int q = 0; q < B; q++)
                    if (j0 + q < U) {
                        if (!okm[n][q]) v[n][q] = c4f(0.0f);
                        acc[n] = acc[n] + c4f(f[j0 + q]) * v[n][q]; /* + f * 0 = + 0: the sum is unchanged, as if the tap were skipped */
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < NI; n++) {
            if (!live[n]) continue;
            c4f a = acc[n];
            if (LAST) {
                if (x[n] + 0 >= X) a.x = 0.0f;
                if (x[n] + 1 >= X) a.y = 0.0f;
                if (x[n] + 2 >= X) a.z = 0.0f;
                if (x[n] + 3 >= X) a.w = 0.0f;
                if (lvl) st4(lvl + at[n], a);
                st4(dog + at[n], ld4(prev + at[n]) - a); /* D = L_prev + (-1) * L_new: an exact subtraction (FeatureIO.cpp:1981) */
            } else
                st4(lvl + at[n], a);
        }
    }
}

template <int R>
__device__ __forceinline__ void level_pass(const chain_ctx &k, int which, const float *in, float *T0, float *T1, float *lvl, float *dog, int XP, int X, int Y,
                           int Z, const float *f)
{
    if (which == 0) pass_x<R>(k, in, T0, XP, (unsigned)Y * (unsigned)Z, f);
    else if (which == 1) pass_col<R, false>(k, T0, T1, nullptr, nullptr, XP, X, Y, Z, false, f);
    else pass_col<R, true>(k, T1, lvl, in, dog, XP, X, Y, Z, true, f);
}

__device__ __forceinline__ void level_pass_any(const chain_ctx &k, int which, int ntaps, const float *taps, const float *in, float *T0, float *T1, float *lvl,
                               float *dog, int XP, int X, int Y, int Z)
{
    float f[2 * SIFT3D_FAST_MAX_R + 1];
#pragma unroll
    for (int j = 0; j < 2 * SIFT3D_FAST_MAX_R + 1; j++) {
        f[j] = j < ntaps ? taps[j] : 0.0f;
        asm volatile("" : "+v"(f[j])); /* in vector registers: as scalars they are spilled and fetched back lane by lane */
    }
    switch (ntaps / 2) {
    case 1: level_pass<1>(k, which, in, T0, T1, lvl, dog, XP, X, Y, Z, f); break;
    case 2: level_pass<2>(k, which, in, T0, T1, lvl, dog, XP, X, Y, Z, f); break;
    case 3: level_pass<3>(k, which, in, T0, T1, lvl, dog, XP, X, Y, Z, f); break;
    case 4: level_pass<4>(k, which, in, T0, T1, lvl, dog, XP, X, Y, Z, f); break;
    case 5: level_pass<5>(k, which, in, T0, T1, lvl, dog, XP, X, Y, Z, f); break;
    case 6: level_pass<6>(k, which, in, T0, T1, lvl, dog, XP, X, Y, Z, f); break;
    case 7: level_pass<7>(k, which, in, T0, T1, lvl, dog, XP, X, Y, Z, f); break;
    default: level_pass<8>(k, which, in, T0, T1, lvl, dog, XP, X, Y, Z, f); break;
    }
}

/* 2 x 2 x 2 mean with the reference's association (FeatureIO.cpp:1532-1538, as subsample_kernel); the pad columns of the
 * coarser octave are written as zeros */
__device__ __forceinline__ void subsample(const chain_ctx &k, const float *__restrict__ in, int XP, int X, int Y, int Z, float *__restrict__ out, int oXP)
{
    const unsigned ox = (unsigned)X / 2u, oy = (unsigned)Y / 2u, oz = (unsigned)Z / 2u;
    const unsigned items = (unsigned)oXP * oy * oz;
    const long long XY = (long long)XP * Y;
    for (unsigned i = k.wg * CHAIN_THREADS + k.tid; i < items; i += k.G * CHAIN_THREADS) {
        const unsigned x = i % (unsigned)oXP;
        const unsigned yz = i / (unsigned)oXP;
        const unsigned y = yz % oy, z = yz / oy;
        float r = 0.0f;
        if (x < ox) {
            const float *p0 = in + (long long)(2 * z) * XY + (long long)(2 * y) * XP + 2 * x;
            const float *p1 = p0 + XY;
            const float a00 = p0[0], a10 = p0[1], a01 = p0[XP], a11 = p0[XP + 1];
            const float b00 = p1[0], b10 = p1[1], b01 = p1[XP], b11 = p1[XP + 1];
            float s = 0.0f;
            s = s + (((a00 + a01) + a10) + a11);
            s = s + (((b00 + b01) + b10) + b11);
            r = s * 0.125f;
        }
        out[i] = r;
    }
}

/* The 26 + 27 + 27 test of one octave's three detection levels on stored DoG levels, a wavefront per 64 x of a row:
 * strictly above (below) the 26 neighbours of its own level, then the 27 of the level below and the 27 of the level
 * above (what extrema_generic_body does per launch). */
__device__ __forceinline__ void extrema(const chain_ctx &k, const sift3d_chain_octave &o, unsigned long long *__restrict__ keys, sift3d_cval *__restrict__ vals,
                        unsigned long long *count, long long cap)
{
    const int X = o.X, XP = o.XP, Y = o.Y, Z = o.Z;
    if (X < 3 || Y < 3 || Z < 3) return;
    const int xb = (X - 2 + 63) / 64;
    const long long XY = (long long)XP * Y;
    const unsigned rows = (unsigned)(Y - 2) * (unsigned)(Z - 2), per_level = rows * (unsigned)xb, items = 3u * per_level;
    const unsigned lane = k.tid & 63u, wave = k.wg * (CHAIN_THREADS / 64u) + (k.tid >> 6), waves = k.G * (CHAIN_THREADS / 64u);
    for (unsigned i = wave; i < items; i += waves) {
        const int l = (int)(i / per_level);
        const unsigned rem = i % per_level;
        const int x = 1 + (int)(rem % (unsigned)xb) * 64 + (int)lane;
        const unsigned yz = rem / (unsigned)xb;
        const int y = 1 + (int)(yz % (unsigned)(Y - 2)), z = 1 + (int)(yz / (unsigned)(Y - 2));
        const float *dprev = o.D[l], *dcur = o.D[l + 1], *dnext = o.D[l + 2];
        const bool inside = x < X - 1;
        const long long idx = (long long)z * XY + (long long)y * XP + x;
        bool mx = inside, mn = inside;
        float c = 0.0f;
        if (inside) c = dcur[idx];
        bool live = true;
#pragma unroll
        for (int dz = -1; dz <= 1; dz++) {
            if (inside && live) {
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        if (dz == 0 && dy == 0 && dx == 0) continue;
                        const float v = dcur[idx + dz * XY + dy * XP + dx];
                        mx = mx && (v < c);
                        mn = mn && (v > c);
                    }
            }
            if (!__any(mx || mn)) live = false;
        }
        if (live && (mx || mn)) {
            const float *lv[2] = {dprev, dnext};
            for (int q = 0; q < 2; q++)
                for (int dz = -1; dz <= 1 && (mx || mn); dz++)
                    for (int dy = -1; dy <= 1; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            const float v = lv[q][idx + dz * XY + dy * XP + dx];
                            mx = mx && (v < c);
                            mn = mn && (v > c);
                        }
            if (mx || mn) {
                const unsigned long long slot = atomicAdd(count, 1ull);
                if ((long long)slot < cap) {
                    sift3d_cval r;
                    r.value = c;
                    r.h = dprev[idx];
                    r.l = dnext[idx];
                    r.pad = 0.0f;
                    keys[slot] = ((unsigned long long)(o.lvl_id0 + l) << SIFT3D_KEY_LVL_SHIFT) | ((unsigned long long)(mx ? 1 : 0) << SIFT3D_KEY_MAX_SHIFT) |
                                 (unsigned long long)idx;
                    vals[slot] = r;
                }
            }
        }
    }
}

/* An octave of at most SIFT3D_TINY_VOX voxels, by one workgroup with the octave in LDS (what tiny_octave_kernel does as a launch
 * of its own): per level x, y, z pass from LDS to LDS, then the level and D = L_prev - L_new to memory.  The caller's barriers
 * order the memory side (the subsample and the extrema test read the levels back). */
template <int AXIS>
__device__ __forceinline__ void lds_pass(const float *src, float *dst, int X, int Y, int Z, int N, const float *f, int nt)
{
    const int R = nt / 2;
    const int len = AXIS == 0 ? X : (AXIS == 1 ? Y : Z);
    const int st = AXIS == 0 ? 1 : (AXIS == 1 ? X : X * Y);
    for (int s = threadIdx.x; s < N; s += CHAIN_THREADS) {
        const int c = AXIS == 0 ? s % X : (AXIS == 1 ? (s / X) % Y : s / (X * Y));
        float acc = 0;
        for (int j = 0; j < nt; j++) {
            const int cc = c + j - R;
            if (cc >= 0 && cc < len) acc = acc + f[j] * src[s + (cc - c) * st];
        }
        dst[s] = acc;
    }
    __syncthreads();
}

__device__ __forceinline__ void octave_in_lds(const chain_ctx &solo, const sift3d_chain_params &p, const sift3d_chain_octave &o,
                                              float (*buf)[SIFT3D_TINY_VOX], float *taps)
{
    const int X = o.X, XP = o.XP, Y = o.Y, Z = o.Z, N = X * Y * Z;
    float *cur = buf[0], *a = buf[1], *b = buf[2];
    for (int s = threadIdx.x; s < N; s += CHAIN_THREADS) cur[s] = o.L[0][(long long)(s / X) * XP + s % X];
    __syncthreads();
    for (int lvl = 0; lvl < 5; lvl++) {
        const int nt = p.ntaps[lvl];
        if ((int)threadIdx.x < nt) taps[threadIdx.x] = p.taps[lvl][threadIdx.x];
        __syncthreads();
        lds_pass<0>(cur, a, X, Y, Z, N, taps, nt);
        lds_pass<1>(a, b, X, Y, Z, N, taps, nt);
        lds_pass<2>(b, a, X, Y, Z, N, taps, nt);
        for (int s = threadIdx.x; s < N; s += CHAIN_THREADS) {
            const long long g = (long long)(s / X) * XP + s % X;
            const float v = a[s];
            if (lvl < 4) o.L[lvl + 1][g] = v;
            o.D[lvl][g] = cur[s] - v;
        }
        /* the pad columns of what was just written hold zeros (nothing but zeros is ever read there): keep them so */
        if (XP != X)
            for (int s = threadIdx.x; s < (XP - X) * Y * Z; s += CHAIN_THREADS) {
                const long long g = (long long)(s / (XP - X)) * XP + X + s % (XP - X);
                if (lvl < 4) o.L[lvl + 1][g] = 0.0f;
                o.D[lvl][g] = 0.0f;
            }
        __syncthreads();
        if (lvl == 2 && o.next_L0) { /* L_3 is in memory: level 0 of the next octave */
            subsample(solo, o.L[3], XP, X, Y, Z, o.next_L0, o.next_XP);
            __syncthreads();
        }
        float *tmp = cur; cur = a; a = tmp;
    }
}

/* Grid-wide barrier: one arrival counter and one generation word (zeroed by the launcher).  The last arriver resets the
 * counter and bumps the generation; the others poll it.  Release before arriving, acquire after leaving (agent scope: the
 * levels one workgroup wrote are read by workgroups on other XCDs, whose L2 is not coherent with this one's without it).
 * Every spin is bounded: a workgroup that waits longer than any healthy run could need raises the abort word, everyone
 * who sees it leaves, and the host reports the failure instead of a hung device. */
__device__ __forceinline__ bool grid_barrier(unsigned *sync, unsigned G, unsigned &gen)
{
    __syncthreads();
    if (G > 1 && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned arrived = __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == G - 1) {
            __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > SIFT3D_CHAIN_SPIN_LIMIT || __hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    gen++;
    __syncthreads();
    if (G > 1) return __hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    return true;
}
} // namespace

__global__ __launch_bounds__(CHAIN_THREADS) void coarse_chain_kernel(sift3d_chain_params p)
{
    __shared__ float tbuf[3][SIFT3D_TINY_VOX];
    __shared__ float ttaps[2 * SIFT3D_FAST_MAX_R + 1];
    chain_ctx k;
    k.wg = blockIdx.x;
    k.G = gridDim.x;
    k.tid = threadIdx.x;
    unsigned gen = 0;
    CHAIN_CLK();
    /* ---- the octaves the whole grid shares: per level x, y, z + DoG; the subsample for the next octave rides with the x
     * pass of level 4 (both only read L_3) ---- */
    for (int oi = 0; oi < p.n_grid; oi++) {
        const sift3d_chain_octave &o = p.oct[oi];
        for (int j = 1; j <= 5; j++) {
            for (int which = 0; which < 3; which++) {
                level_pass_any(k, which, p.ntaps[j - 1], p.taps[j - 1], o.L[j - 1], p.T0, p.T1, j < 5 ? o.L[j] : nullptr, o.D[j - 1], o.XP, o.X,
                               o.Y, o.Z);
                if (j == 4 && which == 0 && o.next_L0) subsample(k, o.L[3], o.XP, o.X, o.Y, o.Z, o.next_L0, o.next_XP);
                CHAIN_CLK();
                if (!grid_barrier(p.sync, k.G, gen)) return;
                CHAIN_CLK();
            }
        }
    }
    if (k.wg == 0) {
        /* ---- workgroup 0 alone: the octaves small enough for one workgroup, with workgroup barriers ---- */
        chain_ctx solo = k;
        solo.G = 1;
        for (int oi = p.n_grid; oi < p.n_oct; oi++) {
            const sift3d_chain_octave &o = p.oct[oi];
            if (o.X * o.Y * o.Z <= SIFT3D_TINY_VOX) {
                octave_in_lds(solo, p, o, tbuf, ttaps);
                CHAIN_CLK();
                extrema(solo, o, p.keys, p.vals, p.count, p.cap);
                CHAIN_CLK();
                continue;
            }
            for (int j = 1; j <= 5; j++)
                for (int which = 0; which < 3; which++) {
                    level_pass_any(solo, which, p.ntaps[j - 1], p.taps[j - 1], o.L[j - 1], p.T0, p.T1, j < 5 ? o.L[j] : nullptr, o.D[j - 1], o.XP,
                                   o.X, o.Y, o.Z);
                    if (j == 4 && which == 0 && o.next_L0) subsample(solo, o.L[3], o.XP, o.X, o.Y, o.Z, o.next_L0, o.next_XP);
                    __syncthreads();
                    CHAIN_CLK();
                }
            extrema(solo, o, p.keys, p.vals, p.count, p.cap);
            CHAIN_CLK();
        }
        if (k.G > 1) return;
    }
    /* ---- everyone else (or the only workgroup): the extrema of the shared octaves ---- */
    chain_ctx e = k;
    if (k.G > 1) {
        e.wg = k.wg - 1;
        e.G = k.G - 1;
    }
    for (int oi = 0; oi < p.n_grid; oi++) extrema(e, p.oct[oi], p.keys, p.vals, p.count, p.cap);
}

hipError_t sift3d_launch_coarse_chain(hipStream_t s, const sift3d_chain_params &p, int workgroups)
{
    if (p.n_oct < 1 || p.n_oct > SIFT3D_CHAIN_MAX_OCT || p.n_grid < 0 || p.n_grid > p.n_oct || workgroups < 1) return hipErrorInvalidValue;
    for (int j = 0; j < 5; j++)
        if (p.ntaps[j] < 3 || p.ntaps[j] > 2 * SIFT3D_FAST_MAX_R + 1 || !(p.ntaps[j] & 1)) return hipErrorNotSupported;
    for (int i = 0; i < p.n_oct; i++)
        if (p.oct[i].XP % 4 != 0 || p.oct[i].X > p.oct[i].XP) return hipErrorNotSupported;
    hipError_t e = hipMemsetAsync(p.sync, 0, sizeof(unsigned) * 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(coarse_chain_kernel, dim3((unsigned)(p.n_grid > 0 ? workgroups : 1)), dim3(CHAIN_THREADS), 0, s, p);
    return hipGetLastError();
}
