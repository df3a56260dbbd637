/*
 * kernels_blur_fused.hip -- the three separable Gaussian passes and the DoG
 * subtraction of one pyramid level in ONE kernel for gfx950 (MI355X).
 *
 * The three-launch path (kernels_volume.hip) moves 32 bytes per voxel and level
 * through HBM (x: 4+4, y: 4+4, z+DoG: 4+4+4+4).  This kernel reads the level's
 * input once and writes the blurred level and the DoG once: 12 bytes per voxel
 * (8 when one of the two outputs is not needed) plus the halo, which neighbouring
 * workgroups share through L2 / Infinity Cache.
 *
 * A workgroup (512 threads, or 256) owns a 64 x 32 (or 64 x 16) (x, y) tile and marches along z:
 *   A  x pass    every thread keeps an aligned window of one input row in
 *                registers (loaded one plane ahead) and produces 8 outputs of
 *                that row; the TY + 2R rows of the tile and its y halo go to one
 *                of two LDS buffers (one barrier per plane);
 *   B  y pass    every thread produces a 2 x 2 block (two rows of one column
 *                pair) from LDS;
 *   C  z pass    the block feeds the thread's 2R+1 partial sums of the output
 *                planes it touches, all in registers (a switch on the plane's
 *                phase makes every slot a compile-time register); the finished
 *                plane is stored together with input - level, the input voxel
 *                (the "previous level" of the DoG) having been loaded a step ahead.
 * The z range is cut into chunks that recompute 2R lead-in planes.  The register
 * file (512 KB per CU) holds what an LDS ring of 2R+1 planes (78 KB per tile for
 * 17 taps) held in the first version of this kernel.
 *
 * Arithmetic contract: identical to kernels_volume.hip (and to the reference's
 * CPU path, R/src_common/GaussBlur3D.cpp:43-61,329-479): every pass is
 *   acc = 0; for j ascending: acc = acc + f[j]*v[j]
 * with separately rounded multiply and add (-ffp-contract=off; the packed
 * v_pk_mul_f32 / v_pk_add_f32 forms are the same IEEE operations on two lanes),
 * zeros outside the volume, float32 between the passes.  A zero tap adds +0 to an
 * accumulator that is never -0, so zero rows / planes and skipped taps agree.
 */
#include <cstdlib>
#include <type_traits>

#include "sift3d_internal.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

/* Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global
 * load (s_waitcnt vmcnt(0)), which would expose the full HBM latency of the next plane's window once per
 * plane; the only data workgroups exchange here lives in LDS. */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#define FB_TX 64
/* planes of window prefetch: two where the registers are there without spilling (13 taps: 243 VGPRs at two wavefronts
 * per SIMD; 17 taps would spill) */
#define FB_PF(R) (((R) == 6 || (R) == 7) ? 2 : 1)

struct fb_taps2 {
    v2f f[2 * SIFT3D_FAST_MAX_R + 1]; /* (f[j], f[j]): a 64-bit scalar operand of the packed multiply */
};

/* TY = rows of the tile (16 or 32), 16 * TY threads: 8 threads per row for the x pass of the TY + 2R rows, a 2 x 2
 * block per thread for the y and z passes.  The taller tile halves the relative cost of the y halo (x pass work and
 * halo traffic) and is used whenever the volume has enough rows. */
template <int R, int TY, int PF>
__global__ __launch_bounds__(16 * TY, (R >= 6 ? 2 : (R == 5 ? 3 : (TY == 32 ? 4 : (R >= 4 ? 3 : 4))))) void blur_fused_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                         float *__restrict__ dog, const float *__restrict__ zeros, int X,
                                                         int Y, int Z, int zlen, int tiles_x, int tiles_y, long long total,
                                                         fb_taps2 t)
{
    constexpr int U = 2 * R + 1;
    constexpr int FB_TY = TY;
    constexpr int FB_P1_ROWS = TY + 2 * SIFT3D_FAST_MAX_R;
    constexpr int NR = FB_TY + 2 * R;        /* rows of the x pass */
    constexpr int H4 = ((R + 3) / 4) * 4;    /* window halo, whole 16-byte vectors */
    constexpr int WIN = 8 + 2 * H4;          /* floats per window */
    constexpr int NV = WIN / 4;              /* vectors per window */
    __shared__ __attribute__((aligned(16))) float P1buf[2][FB_P1_ROWS * FB_TX]; /* double-buffered: one barrier per plane */

    /* workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of tiles (x fastest, then y,
     * then z chunk), i.e. a band of whole rows: the x halos of its tiles and all but the band's two outer y halos
     * are fetched by the L2 that holds the neighbouring tile */
    const long long lin = blockIdx.x;
    const long long per = (total + 7) / 8;
    const long long w = (lin % 8) * per + lin / 8;
    if (w >= total) return;
    const int tx = (int)(w % tiles_x);
    const int ty = (int)((w / tiles_x) % tiles_y);
    const int chunk = (int)(w / ((long long)tiles_y * tiles_x));
    const int x0 = tx * FB_TX, y0 = ty * FB_TY;
    const int zc0 = chunk * zlen;
    const int zc1 = zc0 + zlen < Z ? zc0 + zlen : Z;
    const int tid = threadIdx.x;
    const long long XY = (long long)X * Y;
    const int zfirst = zc0 - R, zlast = zc1 - 1 + R;

    /* stage A role: row ar of the x pass, outputs x0 + axs .. + 7.  Every vector of the window has its own
     * pointer that advances one plane per step; a vector outside the volume points at a zero page and does
     * not advance, so the loop needs no masks. */
    const int ar = tid >> 3, axs = (tid & 7) * 8;
    const int agy = y0 - R + ar;
    const bool arow = ar < NR && agy >= 0 && agy < Y;
    const float *wp[NV]; /* this lane's vector k in the plane being loaded (kept in the global address space) */
    unsigned wstep[NV];  /* floats to the same vector of the next plane (0 on the zero page) */
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int gx = x0 + axs - H4 + 4 * k;
        const bool ok = arow && gx >= 0 && gx < X;
        const long long e = (long long)zfirst * XY + (long long)agy * X + gx; /* negative before plane 0: not loaded */
        wp[k] = ok ? in + e : zeros;
        wstep[k] = ok ? (unsigned)XY : 0u;
    }
    /* stage B/C role: column pair bcp, rows 2*brs and 2*brs+1 of the tile */
    const int bcp = tid & 31, brs = tid >> 5;
    const int bx = x0 + 2 * bcp;
    const int by = y0 + 2 * brs;
    const bool st0 = bx < X && by < Y, st1 = bx < X && by + 1 < Y;
    const long long boff0 = st0 ? (long long)by * X + bx : 0;
    const long long boff1 = st1 ? (long long)(by + 1) * X + bx : 0;

    /* PF = 2: two window buffers; plane z lives in buffer (z - zfirst) & 1 and is loaded two steps before its x pass,
     * so that every wavefront has the loads of two planes in flight (worth its 24 registers where they are free).
     * PF = 1: one buffer, loaded one step ahead. */
    v4f winA[NV], winB[PF == 2 ? NV : 1];
    /* x pass of the plane whose window is in win[]: 8 outputs (4 pairs) to P1.  Output pair e, tap j reads the
     * window floats s, s+1 with s = H4 - R + 2e + j: an aligned register pair when s is even, one of the
     * WIN/2 - 1 odd pairs (built once per plane) when it is odd. */
    auto x_pass = [&](const v4f(&win)[NV], float *P1) {
        v2f ev[WIN / 2], od[WIN / 2 - 1];
#pragma unroll
        for (int k = 0; k < NV; k++) {
            ev[2 * k].x = win[k].x; ev[2 * k].y = win[k].y;
            ev[2 * k + 1].x = win[k].z; ev[2 * k + 1].y = win[k].w;
        }
#pragma unroll
        for (int m = 0; m < WIN / 2 - 1; m++) {
            od[m].x = ev[m].y; od[m].y = ev[m + 1].x;
        }
        v2f o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            v2f acc = v2f(0.0f);
#pragma unroll
            for (int j = 0; j < U; j++) {
                constexpr int base = H4 - R;
                const int sidx = base + 2 * e + j;
                const v2f p = (sidx & 1) ? od[(sidx - 1) / 2] : ev[sidx / 2];
                acc = acc + t.f[j] * p;
            }
            o[e] = acc;
        }
        v4f r0, r1;
        r0.x = o[0].x; r0.y = o[0].y; r0.z = o[1].x; r0.w = o[1].y;
        r1.x = o[2].x; r1.y = o[2].y; r1.z = o[3].x; r1.w = o[3].y;
        *reinterpret_cast<v4f *>(&P1[ar * FB_TX + axs]) = r0;
        *reinterpret_cast<v4f *>(&P1[ar * FB_TX + axs + 4]) = r1;
    };
    /* loads the window of plane z (if it exists) and moves the pointers on to plane z + 1: called for
     * consecutive planes */
    auto load_window = [&](v4f(&win)[NV], int z) {
        if (z >= 0 && z < Z && z <= zlast) {
#pragma unroll
            for (int k = 0; k < NV; k++) win[k] = *reinterpret_cast<const v4f *>(wp[k]);
        }
#pragma unroll
        for (int k = 0; k < NV; k++) wp[k] += wstep[k];
    };
    load_window(winA, zfirst);
    if (zfirst >= 0 && ar < NR) x_pass(winA, P1buf[0]); /* zfirst < Z always */
    if constexpr (PF == 2) {
        load_window(winB, zfirst + 1);
        load_window(winA, zfirst + 2);
    } else {
        load_window(winA, zfirst + 1);
    }

    /* z pass in registers: slot i of acc0/acc1 (rows 2*brs, 2*brs+1) is the output plane that started with the
     * plane of phase i; at phase s the new plane feeds tap (s - i) mod U of slot i, and slot (s + 1) mod U has just
     * received its last tap.  The switch makes every slot index a compile-time register. */
    v2f acc0[U], acc1[U];
#pragma unroll
    for (int i = 0; i < U; i++) acc0[i] = acc1[i] = v2f(0.0f);
    /* the input voxels of the output plane (the previous level of the DoG), loaded one step ahead */
    v2f pv0 = v2f(0.0f), pv1 = v2f(0.0f);
    auto load_prev = [&](int z) {
        if (z < zc1) {
            const float *src = in + (long long)z * XY;
            pv0 = *reinterpret_cast<const v2f *>(src + boff0);
            pv1 = *reinterpret_cast<const v2f *>(src + boff1);
        }
    };
    if (dog) load_prev(zc0);
    int phase = 0, cur = 0;
    lds_barrier(); /* P1buf[0] holds the x pass of plane zfirst */
    auto step = [&](int zin, v4f(&win)[NV]) { /* win holds plane zin + 1 */
        const bool plane = zin >= 0 && zin < Z;
        const int zo = zin - R;
        const bool emit = zo >= zc0; /* zo < zc1 by construction of zlast */
        const float *P1 = P1buf[cur];
        v2f p[U + 1];
#pragma unroll
        for (int q = 0; q < U + 1; q++) p[q] = *reinterpret_cast<const v2f *>(&P1[(2 * brs + q) * FB_TX + 2 * bcp]);
        /* ---- B: y pass of plane zin, 2 rows x 2 columns per thread ---- */
        v2f g0 = v2f(0.0f), g1 = v2f(0.0f);
#pragma unroll
        for (int j = 0; j < U; j++) {
            g0 = g0 + t.f[j] * p[j];
            g1 = g1 + t.f[j] * p[j + 1];
        }
        if (!plane) g0 = g1 = v2f(0.0f); /* P1 was stale */
        /* ---- A: x pass of plane zin + 1 into the other buffer, then the next window for this register buffer ---- */
        if (zin + 1 >= 0 && zin + 1 < Z && zin + 1 <= zlast && ar < NR) x_pass(win, P1buf[cur ^ 1]);
        load_window(win, zin + 1 + PF); /* the buffer is free again */
        /* ---- C: z pass ---- */
        v2f a0 = v2f(0.0f), a1 = v2f(0.0f);
        switch (phase) {
#define FB_PHASE(S)                                                              \
    case S:                                                                      \
        if constexpr (S < U) {                                                   \
            _Pragma("unroll") for (int i = 0; i < U; i++) {                      \
                constexpr int dummy = 0; (void)dummy;                            \
                const int j = (S - i + U) % U;                                   \
                if (j == 0) {                                                    \
                    acc0[i] = v2f(0.0f) + t.f[0] * g0;                           \
                    acc1[i] = v2f(0.0f) + t.f[0] * g1;                           \
                } else {                                                         \
                    acc0[i] = acc0[i] + t.f[j] * g0;                             \
                    acc1[i] = acc1[i] + t.f[j] * g1;                             \
                }                                                                \
            }                                                                    \
            a0 = acc0[(S + 1) % U];                                              \
            a1 = acc1[(S + 1) % U];                                              \
        }                                                                        \
        break;
            FB_PHASE(0) FB_PHASE(1) FB_PHASE(2) FB_PHASE(3) FB_PHASE(4) FB_PHASE(5) FB_PHASE(6) FB_PHASE(7) FB_PHASE(8)
            FB_PHASE(9) FB_PHASE(10) FB_PHASE(11) FB_PHASE(12) FB_PHASE(13) FB_PHASE(14) FB_PHASE(15) FB_PHASE(16)
#undef FB_PHASE
        default: break;
        }
        if (emit) {
            const long long zoff = (long long)zo * XY;
            if (out) {
                if (st0) __builtin_nontemporal_store(a0, reinterpret_cast<v2f *>(out + zoff + boff0));
                if (st1) __builtin_nontemporal_store(a1, reinterpret_cast<v2f *>(out + zoff + boff1));
            }
            if (dog) {
                if (st0) __builtin_nontemporal_store(pv0 - a0, reinterpret_cast<v2f *>(dog + zoff + boff0));
                if (st1) __builtin_nontemporal_store(pv1 - a1, reinterpret_cast<v2f *>(dog + zoff + boff1));
                load_prev(zo + 1); /* for the next step */
            }
        }
        phase = phase + 1 == U ? 0 : phase + 1;
        cur ^= 1;
        lds_barrier(); /* the other buffer is complete, and every wavefront has read this one */
    };
    if constexpr (PF == 2) {
        int zin = zfirst;
        for (; zin + 1 <= zlast; zin += 2) {
            step(zin, winB);
            step(zin + 1, winA);
        }
        if (zin <= zlast) step(zin, winB);
    } else {
        for (int zin = zfirst; zin <= zlast; zin++) step(zin, winA);
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * 11 to 17 taps: the same march with the input planes brought in by LDS-DMA (global_load_lds_dwordx4: global -> LDS, no
 * register destination).  With the window in registers these instantiations have room for one or two planes of prefetch
 * at two wavefronts per SIMD, and the x pass waits for HBM every plane (without its window loads the 13-tap launch takes
 * 0.40 ms instead of 0.59).  Here a ring of NBUF raw planes (the tile's TY + 2R rows of 64 + 16 floats) lives in LDS,
 * two planes in flight, at the cost of no registers (which lets two workgroups share a CU); the input voxel of the DoG comes the same way (so that no ordinary
 * global load is left in the loop: beside LDS-DMA the compiler would wait for vmcnt(0) at each use of one).
 *
 * LDS-DMA writes base + lane * 16 bytes: a plane is dealt to the wavefronts in groups of 64 consecutive 16-byte vectors
 * (row-major, 20 vectors per row), GPW groups per wavefront; lanes past the last vector and vectors outside the volume
 * read a zero page.  Ordering: a wavefront's DMA is covered by its own counted s_waitcnt vmcnt(N) (vector memory
 * operations retire in issue order) and, for the other wavefronts' reads, by the barrier that ends the step; every step
 * issues the same number of DMA operations so that N is a constant.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef __attribute__((address_space(1))) const void fb_gptr;
typedef __attribute__((address_space(3))) void fb_lptr;

/* NBUF: raw planes in the ring, one being read and NBUF - 1 in flight.  3: 80 KB of LDS, two workgroups per CU where the
 * kernel stays within 128 registers; 4: 110 KB, one workgroup per CU with a deeper prefetch. */
template <int R, int TY, int NBUF>
__global__ __launch_bounds__(16 * TY, (NBUF == 3 ? 4 : 2)) void blur_fused_dma_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                                    float *__restrict__ dog, const float *__restrict__ zeros,
                                                                    float *__restrict__ sink, int X, int Y, int Z, int zlen,
                                                                    int tiles_x, int tiles_y, long long total, fb_taps2 t)
{
    static_assert(R >= 5 && R <= 8 && TY == 32, "window halo of 8 floats, eight wavefronts");
    constexpr int PD = NBUF - 2;  /* steps between the issue of a DMA and the wait that covers it */
    constexpr int NPV = PD + 1;   /* DoG-input planes in LDS: issued PD steps before their use */
    constexpr int U = 2 * R + 1;
    constexpr int NR = TY + 2 * R;           /* rows of the x pass */
    constexpr int H4 = 8;                    /* window halo, whole 16-byte vectors */
    constexpr int ROWF = FB_TX + 2 * H4;     /* floats of a raw row */
    constexpr int ROWV = ROWF / 4;           /* vectors of a raw row */
    constexpr int NVEC = NR * ROWV;          /* vectors of a raw plane */
    constexpr int NW = TY / 4;               /* wavefronts */
    constexpr int GR = (NVEC + 63) / 64;     /* DMA groups of a plane: wavefront w takes group w and, if it exists, w + NW */
    static_assert(GR > NW && GR <= 2 * NW, "one or two groups per wavefront");
    constexpr int PLANE = GR * 256;          /* floats of a ring slot */
    constexpr int PVPL = TY * FB_TX;         /* floats of a DoG-input plane: exactly one DMA group per wavefront */
    constexpr int P1PL = NR * FB_TX;
    constexpr int WIN = 8 + 2 * H4, NV = WIN / 4;
    /* one array: a second __shared__ object beside an LDS-DMA target makes the compiler drain vmcnt before LDS reads */
    __shared__ __attribute__((aligned(1024))) float lds[NBUF * PLANE + NPV * PVPL + 2 * P1PL];
    float *const raw = lds;
    float *const pvb = lds + NBUF * PLANE;
    float *const P1b = pvb + NPV * PVPL;

    const long long lin = blockIdx.x;
    const long long per = (total + 7) / 8;
    const long long wi = (lin % 8) * per + lin / 8;
    if (wi >= total) return;
    const int tx = (int)(wi % tiles_x);
    const int ty = (int)((wi / tiles_x) % tiles_y);
    const int chunk = (int)(wi / ((long long)tiles_y * tiles_x));
    const int x0 = tx * FB_TX, y0 = ty * TY;
    const int zc0 = chunk * zlen;
    const int zc1 = zc0 + zlen < Z ? zc0 + zlen : Z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long XY = (long long)X * Y;
    const int zfirst = zc0 - R, zlast = zc1 - 1 + R;

    /* DMA roles.  Raw plane: vector v = 64 * (wv + NW * k) + lane -> row v / ROWV, vector v % ROWV of the row. */
    constexpr int GPW = 2;
    const int ngr = wv + NW < GR ? 2 : 1; /* groups of this wavefront */
    int roff[GPW]; /* this lane's vector k: element offset inside a plane (X * Y < 2^29), -1 = outside the volume */
#pragma unroll
    for (int k = 0; k < GPW; k++) {
        const int v = 64 * (wv + NW * k) + lane;
        const int row = v / ROWV, col = v - row * ROWV;
        const int gy = y0 - R + row, gx = x0 - H4 + 4 * col;
        roff[k] = (v < NVEC && gy >= 0 && gy < Y && gx >= 0 && gx < X) ? gy * X + gx : -1;
    }
    /* DoG input plane: vector 64 * wv + lane -> row (64 * wv + lane) / 16: the rows this wavefront's own threads take */
    const int pvr = (64 * wv + lane) >> 4, pvc = (64 * wv + lane) & 15;
    const int poff = (y0 + pvr < Y && x0 + 4 * pvc < X) ? (y0 + pvr) * X + x0 + 4 * pvc : -1;
    auto issue_raw = [&](int z, int slot) { /* plane z -> ring slot; wave-uniform z */
        const bool zin_vol = z >= 0 && z < Z && z <= zlast;
#pragma unroll
        for (int k = 0; k < GPW; k++) {
            if (k < ngr) { /* wave-uniform */
                const float *src = (zin_vol && roff[k] >= 0) ? in + ((long long)z * XY + roff[k]) : zeros;
                __builtin_amdgcn_global_load_lds((fb_gptr *)src, (fb_lptr *)(raw + slot * PLANE + (wv + NW * k) * 256), 16, 0, 0);
            }
        }
    };
    auto issue_prev = [&](int z, int slot) {
        const float *src = (z >= zc0 && z < zc1 && poff >= 0 && dog) ? in + ((long long)z * XY + poff) : zeros;
        __builtin_amdgcn_global_load_lds((fb_gptr *)src, (fb_lptr *)(pvb + slot * PVPL + wv * 256), 16, 0, 0);
    };

    /* stage A role: row ar of the x pass, outputs x0 + axs .. + 7, window = floats axs .. axs + 23 of the raw row */
    const int ar = tid >> 3, axs = (tid & 7) * 8;
    auto x_pass = [&](const float *slotp, float *P1) {
        v4f win[NV];
        const v4f *rw = reinterpret_cast<const v4f *>(slotp + ar * ROWF + axs);
#pragma unroll
        for (int k = 0; k < NV; k++) win[k] = rw[k];
        v2f ev[WIN / 2], od[WIN / 2 - 1];
#pragma unroll
        for (int k = 0; k < NV; k++) {
            ev[2 * k].x = win[k].x; ev[2 * k].y = win[k].y;
            ev[2 * k + 1].x = win[k].z; ev[2 * k + 1].y = win[k].w;
        }
#pragma unroll
        for (int m = 0; m < WIN / 2 - 1; m++) {
            od[m].x = ev[m].y; od[m].y = ev[m + 1].x;
        }
        v2f o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            v2f acc = v2f(0.0f);
#pragma unroll
            for (int j = 0; j < U; j++) {
                constexpr int base = H4 - R;
                const int sidx = base + 2 * e + j;
                const v2f p = (sidx & 1) ? od[(sidx - 1) / 2] : ev[sidx / 2];
                acc = acc + t.f[j] * p;
            }
            o[e] = acc;
        }
        v4f r0, r1;
        r0.x = o[0].x; r0.y = o[0].y; r0.z = o[1].x; r0.w = o[1].y;
        r1.x = o[2].x; r1.y = o[2].y; r1.z = o[3].x; r1.w = o[3].y;
        *reinterpret_cast<v4f *>(&P1[ar * FB_TX + axs]) = r0;
        *reinterpret_cast<v4f *>(&P1[ar * FB_TX + axs + 4]) = r1;
    };
    /* stage B/C role: column pair bcp, rows 2*brs and 2*brs+1 of the tile */
    const int bcp = tid & 31, brs = tid >> 5;
    const int bx = x0 + 2 * bcp;
    const int by = y0 + 2 * brs;
    const int boff0 = (bx < X && by < Y) ? by * X + bx : -1;         /* element offset inside a plane, -1: store to the sink */
    const int boff1 = (bx < X && by + 1 < Y) ? (by + 1) * X + bx : -1;

    /* prologue: planes zfirst .. zfirst + NBUF - 1 into slots 0 .. NBUF - 1, x pass of plane zfirst */
#pragma unroll
    for (int b = 0; b < NBUF; b++) issue_raw(zfirst + b, b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (zfirst >= 0 && ar < NR) x_pass(raw, P1b); /* zfirst < Z always */
    v2f acc0[U], acc1[U];
#pragma unroll
    for (int i = 0; i < U; i++) acc0[i] = acc1[i] = v2f(0.0f);
    int phase = 0, cur = 0;
    int xslot = 1;  /* ring slot of plane zin + 1 */
    const int ns = (out ? 2 : 0) + (dog ? 2 : 0); /* stores per storing step */
    int pslot_w = 0, pslot_r = 0; /* DoG-input ring: slot filled this step (plane zin + PD - R), slot read (plane zin - R) */
    lds_barrier(); /* P1b[0] holds the x pass of plane zfirst, and every wavefront has read slot 0 */
    for (int zin = zfirst; zin <= zlast; zin++) {
        const bool plane = zin >= 0 && zin < Z;
        const int zo = zin - R;
        const bool emit = zo >= zc0; /* zo < zc1 by construction of zlast */
        const float *P1 = P1b + cur * P1PL;
        v2f p[U + 1];
#pragma unroll
        for (int q = 0; q < U + 1; q++) p[q] = *reinterpret_cast<const v2f *>(&P1[(2 * brs + q) * FB_TX + 2 * bcp]);
        /* ---- B: y pass of plane zin ---- */
        v2f g0 = v2f(0.0f), g1 = v2f(0.0f);
#pragma unroll
        for (int j = 0; j < U; j++) {
            g0 = g0 + t.f[j] * p[j];
            g1 = g1 + t.f[j] * p[j + 1];
        }
        if (!plane) g0 = g1 = v2f(0.0f); /* P1 was stale */
        /* ---- A: x pass of plane zin + 1 from its ring slot into the other P1 buffer ---- */
        if (zin + 1 >= 0 && zin + 1 < Z && zin + 1 <= zlast && ar < NR) x_pass(raw + xslot * PLANE, P1b + (cur ^ 1) * P1PL);
        /* ---- DMA: the DoG input of the plane stored two steps from now, then plane zin + NBUF into the slot plane
         * zin left (every wavefront passed the barrier after reading it).  Always issued (a zero page when there is
         * nothing to fetch), so that the count below is exact. ---- */
        issue_prev(zo + PD, pslot_w);
        issue_raw(zin + NBUF, xslot == 0 ? NBUF - 1 : xslot - 1);
        /* Wait for the DMA issued PD steps ago: plane zin + 2 (read next step) and the DoG input read below.  Younger
         * than those are the 1 + ngr DMA operations of each step since and the stores of the last PD steps: exactly `ns`
         * per step once the march stores (lanes outside the volume store to a sink instead of being skipped).  Vector
         * memory operations retire in issue order, so "all but the N youngest are done" covers them. */
        {
            int nyoung = PD * (1 + ngr);
#pragma unroll
            for (int k = 1; k <= PD; k++) nyoung += (zo - k >= zc0) ? ns : 0;
            switch (nyoung) {
#define FB_WAIT(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
                FB_WAIT(2) FB_WAIT(3) FB_WAIT(4) FB_WAIT(5) FB_WAIT(6) FB_WAIT(7) FB_WAIT(8) FB_WAIT(9) FB_WAIT(10)
                FB_WAIT(11) FB_WAIT(12) FB_WAIT(13) FB_WAIT(14)
#undef FB_WAIT
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        /* ---- C: z pass ---- */
        v2f a0 = v2f(0.0f), a1 = v2f(0.0f);
        switch (phase) {
#define FB_PHASE(S)                                                              \
    case S:                                                                      \
        if constexpr (S < U) {                                                   \
            _Pragma("unroll") for (int i = 0; i < U; i++) {                      \
                const int j = (S - i + U) % U;                                   \
                if (j == 0) {                                                    \
                    acc0[i] = v2f(0.0f) + t.f[0] * g0;                           \
                    acc1[i] = v2f(0.0f) + t.f[0] * g1;                           \
                } else {                                                         \
                    acc0[i] = acc0[i] + t.f[j] * g0;                             \
                    acc1[i] = acc1[i] + t.f[j] * g1;                             \
                }                                                                \
            }                                                                    \
            a0 = acc0[(S + 1) % U];                                              \
            a1 = acc1[(S + 1) % U];                                              \
        }                                                                        \
        break;
            FB_PHASE(0) FB_PHASE(1) FB_PHASE(2) FB_PHASE(3) FB_PHASE(4) FB_PHASE(5) FB_PHASE(6) FB_PHASE(7) FB_PHASE(8)
            FB_PHASE(9) FB_PHASE(10) FB_PHASE(11) FB_PHASE(12) FB_PHASE(13) FB_PHASE(14) FB_PHASE(15) FB_PHASE(16)
#undef FB_PHASE
        default: break;
        }
        if (emit) { /* wave-uniform; every lane stores (see the wait above) */
            const long long zoff = (long long)zo * XY;
            if (out) {
                __builtin_nontemporal_store(a0, reinterpret_cast<v2f *>(boff0 >= 0 ? out + (zoff + boff0) : sink));
                __builtin_nontemporal_store(a1, reinterpret_cast<v2f *>(boff1 >= 0 ? out + (zoff + boff1) : sink));
            }
            if (dog) {
                const float *pvp = pvb + pslot_r * PVPL + (2 * brs) * FB_TX + 2 * bcp;
                const v2f pv0 = *reinterpret_cast<const v2f *>(pvp);
                const v2f pv1 = *reinterpret_cast<const v2f *>(pvp + FB_TX);
                __builtin_nontemporal_store(pv0 - a0, reinterpret_cast<v2f *>(boff0 >= 0 ? dog + (zoff + boff0) : sink));
                __builtin_nontemporal_store(pv1 - a1, reinterpret_cast<v2f *>(boff1 >= 0 ? dog + (zoff + boff1) : sink));
            }
        }
        phase = phase + 1 == U ? 0 : phase + 1;
        cur ^= 1;
        xslot = xslot + 1 == NBUF ? 0 : xslot + 1;
        pslot_w = pslot_w + 1 == NPV ? 0 : pslot_w + 1;
        if (zin - zfirst >= PD) pslot_r = pslot_r + 1 == NPV ? 0 : pslot_r + 1; /* read slot = written PD steps before */
        lds_barrier(); /* the other P1 buffer is complete, every wavefront has read this one and its ring slot */
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* no DMA may land after the workgroup has released its LDS */
}

/* ------------------------------------------------------------------------------------------------------------------
 * Second form of the register-window march ("ring" kernel), round 2.  Same tile, same three passes, same arithmetic;
 * what changed is everything around the arithmetic:
 *
 *  - The DoG's input voxel ("previous level") is no longer re-read from memory R planes after the window load: the
 *    x-pass thread already holds it (the centre 8 floats of its window) and drops it into an LDS ring of R+2 planes of
 *    the tile (8 KB each), from which the thread that stores the DoG picks it up R steps later.  Read side: 4 B/voxel
 *    plus halo instead of 8.
 *  - Every global access is a buffer operation (buffer_load_dwordx4 / buffer_store_dwordx2 ... offen, plane offset in an
 *    SGPR): a lane outside the volume carries the offset 0xFFFFFFFF and the hardware's bounds check returns zeros / drops
 *    the store; a plane outside the volume uses a descriptor of zero records.  No pointer arithmetic in vector registers,
 *    no zero page, no exec-mask branches around loads and stores -- so the loop body is straight-line code with a fixed
 *    number of vector-memory operations, and the wait in front of the x pass is "all but this step's stores"
 *    (s_waitcnt vmcnt(#stores)) instead of vmcnt(0).  The first form drained its own stores once per plane: with loads
 *    or stores alone it took 0.21 / 0.24 ms at 512^3 (7 taps), with both 0.40.
 *  - Lead-in steps (no output plane yet) issue their stores too, through a descriptor of zero records (dropped by the
 *    bounds check), so that every step and every path into the loop carries the same vmcnt state.
 *
 * BR = output rows per thread: 2 (512 threads, 2 x 2 block, the first form's mapping) or 1 (1024 threads, 2 x 1 block,
 * half the accumulator registers per thread: sixteen wavefronts per workgroup for the wide filters, which would
 * otherwise sit at two wavefronts per SIMD).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
#define FB_RSRC_FLAGS 0x00020000 /* raw buffer, 32-bit data format (the value composable_kernel uses for gfx90a..gfx950) */
#define FB_OOB 0xFFFFFFFFu        /* >= any num_records: the lane's access is out of range by construction */

/* XO = outputs per lane of the x pass: 8 (a wavefront filters 8 rows of the tile) or 4 (4 rows).  With 8, the TY + 2R rows
 * are 5 or 6 wavefronts' worth, which sixteen wavefronts on four SIMDs cannot share evenly (two SIMDs carry two x-pass
 * wavefronts, two carry one: 512 against 376 packed operations per plane at 17 taps); with 4 they are 10 to 12 wavefronts,
 * three per SIMD.  Only the 1024-thread mapping has the wavefronts for that. */
template <int R, int BR, int PF = 1, int XO = 8>
struct fb_ring_cfg {
    static constexpr int TY = 32;
    static constexpr int NT = 1024 / BR;
    static constexpr int NR = TY + 2 * R;
    static constexpr int LPR = FB_TX / XO;                      /* lanes per row of the x pass */
    static constexpr int RPW = 64 / LPR;                        /* rows per x-pass wavefront */
    static constexpr int XW = (NR + RPW - 1) / RPW;             /* wavefronts with an x-pass role */
    static constexpr int P1ROWS = XW * RPW;                     /* rows those wavefronts write (>= NR; the surplus rows are never read) */
    static constexpr int S = R + 2;                /* DoG-input ring: written for plane z+1 while plane z-R is read */
    /* SINGLE: the configuration meant to run one workgroup per CU (two rows per thread, two planes of prefetch).  Its
     * LDS request is padded past half of the CU's 160 KiB so that a second workgroup can never become resident, also on
     * volumes with more tiles than CUs: two resident workgroups with two planes in flight each stream markedly slower
     * (0.38 against 0.315 ms per 7-tap launch at 512^3). */
    static constexpr bool SINGLE = PF == 2 && BR == 2;
    static constexpr int LDS_NEEDED = 2 * P1ROWS * FB_TX + S * TY * FB_TX;
    static constexpr int LDS_FLOATS = (SINGLE && LDS_NEEDED < 21 * 1024) ? 21 * 1024 : LDS_NEEDED;
    /* workgroups per CU the LDS allows (160 KiB), capped at what 32 wavefronts per CU allow */
    static constexpr int WG_LDS = (160 * 1024) / (LDS_FLOATS * 4);
    static constexpr int WG_WAVES = 32 / (NT / 64);
    static constexpr int WG = WG_LDS < WG_WAVES ? (WG_LDS < 1 ? 1 : WG_LDS) : WG_WAVES;
    static constexpr int WAVES_PER_SIMD = WG * (NT / 64) / 4 > 4 ? 4 : WG * (NT / 64) / 4; /* never ask for fewer than 128 registers */
};

/* Register budget: with two planes of prefetch the 512-thread mapping is meant to run ONE workgroup per CU (two
 * wavefronts per SIMD, up to 256 registers) -- a zero-arithmetic march of the same tiles streams 5.4-5.8 TB/s with one
 * workgroup per CU and two z chunks against 4.4-4.9 with two per CU and four chunks (tools/stream_roof.hip), and the
 * second plane in flight covers the latency the second workgroup covered. */
template <int R, int BR, bool HAS_OUT, bool HAS_DOG, int PF, int XO>
__global__ __launch_bounds__(1024 / BR, (fb_ring_cfg<R, BR, PF, XO>::WAVES_PER_SIMD)) void blur_fused_ring_kernel(
    const float *__restrict__ in, float *__restrict__ out, float *__restrict__ dog, int X, int Y, int Z, int zlen, int tiles_x,
    int tiles_y, long long total, int prio, fb_taps2 t)
{
    using C = fb_ring_cfg<R, BR, PF, XO>;
    constexpr int U = 2 * R + 1;
    constexpr int TY = C::TY, NR = C::NR, XW = C::XW, P1ROWS = C::P1ROWS, S = C::S;
    constexpr int H4 = ((R + 3) / 4) * 4; /* window halo, whole 16-byte vectors */
    constexpr int WIN = XO + 2 * H4, NV = WIN / 4;
    constexpr int LPR = C::LPR;
    static_assert(XO == 8 || XO == 4, "a lane filters one or two 16-byte vectors of a row");
    constexpr int P1PL = P1ROWS * FB_TX, PVPL = TY * FB_TX;
    static_assert(XW * 64 <= C::NT, "the x-pass wavefronts are wavefronts of the workgroup");
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    float *const P1b = lds;
    float *const pvb = lds + 2 * P1PL;

    const long long lin = blockIdx.x;
    const long long per = (total + 7) / 8;
    const long long wi = (lin % 8) * per + lin / 8; /* XCD-aware tile order, as in the first form */
    if (wi >= total) return;
    const int tx = (int)(wi % tiles_x);
    const int ty = (int)((wi / tiles_x) % tiles_y);
    const int chunk = (int)(wi / ((long long)tiles_y * tiles_x));
    const int x0 = tx * FB_TX, y0 = ty * TY;
    const int zc0 = chunk * zlen;
    const int zc1 = zc0 + zlen < Z ? zc0 + zlen : Z;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long XY = (long long)X * Y;
    const unsigned plane_bytes = (unsigned)(XY * 4); /* X * Y < 2^29 */
    const int zfirst = zc0 - R, zlast = zc1 - 1 + R;
    const int zb = zfirst > 0 ? zfirst : 0; /* first plane the input descriptor covers */

    /* Buffer descriptors.  Offsets are 32-bit: the launcher keeps (zlen + 2R + 2) planes below 4 GiB, every descriptor
     * starts at this chunk's first plane, and its record count ends at the end of the volume. */
    auto nrec = [&](int zfrom) -> unsigned {
        const long long b = (long long)(Z - zfrom) * XY * 4;
        return b > 0xFFFFFFF0ll ? 0xFFFFFFF0u : (unsigned)b;
    };
    const unsigned in_rec = nrec(zb);
    const float *const in_base = in + (long long)zb * XY;
    float *const out_base = HAS_OUT ? out + (long long)zc0 * XY : nullptr;
    float *const dog_base = HAS_DOG ? dog + (long long)zc0 * XY : nullptr;
    const unsigned st_rec = nrec(zc0);
    /* The stores of a step are issued on EVERY step, lead-in steps included, through a descriptor whose record count is
     * zero until the first output plane is complete: the bounds check drops them, but they count in vmcnt like real
     * ones.  That keeps the number of vector-memory operations per step constant on every path into and around the
     * loop, which is what lets the compiler wait with vmcnt(#stores + k) in front of the x pass instead of vmcnt(0). */
    auto store_rsrc = [&](float *base, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, live ? (int)st_rec : 0, FB_RSRC_FLAGS);
    };

    /* stage A role (wavefronts 0 .. XW-1): row ar of the x pass, outputs x0 + axs .. + XO-1.  Lanes past row NR-1 and rows
     * outside the volume carry out-of-range offsets: they filter zeros into P1 rows nobody reads / rows that are zero. */
    const bool xrole = wv < XW; /* wave-uniform */
    const int ar = tid / LPR, axs = (tid % LPR) * XO;
    const int agy = y0 - R + ar;
    const bool arow = ar < NR && agy >= 0 && agy < Y;
    unsigned voff[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int gx = x0 + axs - H4 + 4 * k;
        voff[k] = (arow && gx >= 0 && gx < X) ? (unsigned)(agy * X + gx) * 4u : FB_OOB;
    }
    const bool prow = ar >= R && ar < R + TY; /* a row of the tile itself: its centre 8 floats go to the DoG-input ring */

    /* stage B/C role: column pair bcp, BR rows starting at row brow of the tile */
    const int bcp = tid & 31, brow = (tid >> 5) * BR;
    const int bx = x0 + 2 * bcp;
    unsigned soff[BR];
#pragma unroll
    for (int r = 0; r < BR; r++) soff[r] = (bx < X && y0 + brow + r < Y) ? (unsigned)((y0 + brow + r) * X + bx) * 4u : FB_OOB;

    /* PF planes of window prefetch: plane z lives in buffer (z - zfirst) % PF and is loaded PF steps before its x pass */
    v4f win[PF][NV];
    auto load_window = [&](int z, auto bsel) {
        constexpr int B = decltype(bsel)::value; /* wave-uniform z; a plane outside the volume reads through a descriptor of no records */
        const bool ok = z >= 0 && z < Z;
        const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc((void *)in_base, 0, ok ? (int)in_rec : 0, FB_RSRC_FLAGS);
        const unsigned so = ok ? (unsigned)(z - zb) * plane_bytes : 0u;
#pragma unroll
        for (int k = 0; k < NV; k++) win[B][k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)voff[k], (int)so, 0));
    };
    /* x pass of the plane in win[] into P1, centre of the window into the DoG-input ring slot pvs */
    auto x_pass = [&](float *P1, float *pvs, auto bsel) {
        constexpr int B = decltype(bsel)::value;
        /* Not every float of the window is a filter input (7 taps: floats 1..14 of 16).  Left to itself the register
         * allocator hands the dead element of a loaded vector to a temporary right after the load is issued, and the
         * write-after-write hazard on the in-flight load costs an s_waitcnt vmcnt(0) there, one plane early.  The empty
         * asm makes all four elements live until this point, where the window is consumed anyway. */
#pragma unroll
        for (int k = 0; k < NV; k++) asm volatile("" : "+v"(win[B][k]));
        v2f ev[WIN / 2], od[WIN / 2 - 1];
#pragma unroll
        for (int k = 0; k < NV; k++) {
            ev[2 * k].x = win[B][k].x; ev[2 * k].y = win[B][k].y;
            ev[2 * k + 1].x = win[B][k].z; ev[2 * k + 1].y = win[B][k].w;
        }
#pragma unroll
        for (int m = 0; m < WIN / 2 - 1; m++) {
            od[m].x = ev[m].y; od[m].y = ev[m + 1].x;
        }
        if (HAS_DOG && prow) {
#pragma unroll
            for (int q = 0; q < XO / 4; q++) *reinterpret_cast<v4f *>(&pvs[(ar - R) * FB_TX + axs + 4 * q]) = win[B][H4 / 4 + q];
        }
        /* tap loop outside, output pairs inside: per tap XO/2 independent products, then XO/2 adds, none of which depends
         * on its immediate predecessor (a packed operation that consumes the result of the instruction right before it
         * costs a wait state: with the pair loop outside the compiler issued 41 s_nop per 136 packed operations at 17
         * taps).  Every accumulator still receives its products in ascending tap order. */
        /* (one accumulator chain per output pair; issuing the taps outermost instead -- XO/2 independent products, then
         * XO/2 adds -- removes a fifth of the s_nop the compiler places between dependent packed operations and changes
         * nothing measurable: the other wavefronts of the SIMD fill those slots) */
        v2f o[XO / 2];
#pragma unroll
        for (int e = 0; e < XO / 2; e++) {
            v2f acc = v2f(0.0f);
#pragma unroll
            for (int j = 0; j < U; j++) {
                constexpr int base = H4 - R;
                const int sidx = base + 2 * e + j;
                const v2f p = (sidx & 1) ? od[(sidx - 1) / 2] : ev[sidx / 2];
                acc = acc + t.f[j] * p;
            }
            o[e] = acc;
        }
#pragma unroll
        for (int q = 0; q < XO / 4; q++) {
            v4f r0;
            r0.x = o[2 * q].x; r0.y = o[2 * q].y; r0.z = o[2 * q + 1].x; r0.w = o[2 * q + 1].y;
            *reinterpret_cast<v4f *>(&P1[ar * FB_TX + axs + 4 * q]) = r0;
        }
    };

    auto store_plane = [&](bool live, unsigned so, const v2f(&lv)[BR], const v2f(&dg)[BR]) {
        if constexpr (HAS_OUT) {
            const __amdgpu_buffer_rsrc_t r = store_rsrc(out_base, live);
#pragma unroll
            for (int q = 0; q < BR; q++) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, lv[q]), r, (int)soff[q], (int)so, 2 /* nt */);
        }
        if constexpr (HAS_DOG) {
            const __amdgpu_buffer_rsrc_t r = store_rsrc(dog_base, live);
#pragma unroll
            for (int q = 0; q < BR; q++) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, dg[q]), r, (int)soff[q], (int)so, 2 /* nt */);
        }
    };

    v2f acc[BR][U];
#pragma unroll
    for (int r = 0; r < BR; r++)
#pragma unroll
        for (int i = 0; i < U; i++) acc[r][i] = v2f(0.0f);
    int cur = 0;
    int wslot = 0; /* ring slot of the plane the x pass is working on; the plane stored this step sits in wslot + 1 (mod R+2) */

    /* The x-pass wavefronts carry the longest step (x + y + z pass against y + z) and everyone meets them at the barrier:
     * give them issue priority over the wavefronts that share their SIMD (`prio`: 0 leaves the arbitration to age). */
    if (xrole && prio) __builtin_amdgcn_s_setprio(2);
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, PF - 1>;
    if (xrole) {
        load_window(zfirst, B0{});
        x_pass(P1b, pvb, B0{});        /* plane zfirst (zeros when it lies before the volume) */
        if constexpr (PF == 2) {
            load_window(zfirst + 1, B1{});
            load_window(zfirst + 2, B0{});
        } else {
            load_window(zfirst + 1, B0{});
        }
    }
    {
        v2f z2[BR];
#pragma unroll
        for (int r = 0; r < BR; r++) z2[r] = v2f(0.0f);
        store_plane(false, 0u, z2, z2); /* dropped; same vmcnt state as the loop's back edge */
    }
    wslot = 1;
    lds_barrier();

    /* one step: y pass of plane zin from P1[cur], x pass of plane zin + 1 into the other buffer, z pass; EMIT: plane
     * zin - R is complete and is stored */
    auto step = [&](int zin, auto bsel) { /* bsel: the window buffer that holds plane zin + 1 */
        const bool emit = zin - R >= zc0; /* wave-uniform: plane zin - R is complete (it is < zc1 by construction of zlast) */
        const float *P1 = P1b + cur * P1PL;
        v2f p[U + BR - 1];
#pragma unroll
        for (int q = 0; q < U + BR - 1; q++) p[q] = *reinterpret_cast<const v2f *>(&P1[(brow + q) * FB_TX + 2 * bcp]);
        v2f g[BR];
#pragma unroll
        for (int r = 0; r < BR; r++) {
            v2f a = v2f(0.0f);
#pragma unroll
            for (int j = 0; j < U; j++) a = a + t.f[j] * p[j + r];
            g[r] = a;
        }
        if (xrole) {
            x_pass(P1b + (cur ^ 1) * P1PL, pvb + wslot * PVPL, bsel);
            load_window(zin + 1 + PF, bsel); /* the buffer is free again */
        }
        /* ---- C: z pass, shift form.  Slot i holds the output plane that completes in i + 1 steps and therefore takes
         * tap U-1-i of the new plane; the sum moves one slot down as it is updated (a three-operand add reads slot i+1 and
         * writes slot i, so the shift costs nothing), slot U-1 restarts from 0 + f[0]*g.  Every slot index is a
         * compile-time register without a switch on the plane's phase: the first form's switch cost 2R+1 register
         * copies per step (the phi nodes of its cases), 2R+1 copies of the z-pass code and a basic-block boundary
         * between the passes.  The taps are bit-symmetric (checked by the launcher), so f[j]*g and f[U-1-j]*g are
         * one product: R+1 multiplies instead of 2R+1, the additions and their order unchanged. ---- */
        v2f a[BR];
#pragma unroll
        for (int r = 0; r < BR; r++) {
            v2f prod[R + 1];
#pragma unroll
            for (int k = 0; k <= R; k++) prod[k] = t.f[k] * g[r];
#pragma unroll
            for (int i = 0; i < U; i++) {
                const int k = i < U - 1 - i ? i : U - 1 - i;
                if (i + 1 < U) acc[r][i] = acc[r][i + 1] + prod[k];
                else acc[r][i] = v2f(0.0f) + prod[k];
            }
            a[r] = acc[r][0];
        }
        {
            const unsigned so = emit ? (unsigned)(zin - R - zc0) * plane_bytes : 0u;
            v2f dg[BR];
            if constexpr (HAS_DOG) {
                const int rslot = wslot + 1 == S ? 0 : wslot + 1;
                const float *pvp = pvb + rslot * PVPL + brow * FB_TX + 2 * bcp;
#pragma unroll
                for (int r = 0; r < BR; r++) dg[r] = *reinterpret_cast<const v2f *>(pvp + r * FB_TX) - a[r];
            } else {
#pragma unroll
                for (int r = 0; r < BR; r++) dg[r] = v2f(0.0f);
            }
            store_plane(emit, so, a, dg);
        }
        cur ^= 1;
        wslot = wslot + 1 == S ? 0 : wslot + 1;
        lds_barrier(); /* the other P1 buffer and the ring slot are complete, every wavefront has read this P1 buffer */
    };
    /* the first 2R steps are lead-in: their stores are dropped */
    if constexpr (PF == 2) {
        int zin = zfirst;
        for (; zin + 1 <= zlast; zin += 2) {
            step(zin, B1{});     /* plane zfirst + 1 went to buffer 1 */
            step(zin + 1, B0{});
        }
        if (zin <= zlast) step(zin, B1{});
    } else {
        for (int zin = zfirst; zin <= zlast; zin++) step(zin, B0{});
    }
}

/* chunks along z: enough workgroups to fill every CU's resident slots while the 2R lead-in planes stay cheap */
static int fused_chunks(int R, int64_t Z, long long tiles, int resident)
{
    const char *env = getenv("SIFT3D_FUSED_CHUNKS"); /* tuning / test aid: force the number of z chunks */
    if (env && atoi(env) >= 1) return atoi(env);
    /* time ~ rounds of resident workgroups x planes marched per workgroup */
    const double slots = 256.0 * resident;
    int best = 1;
    double best_cost = 0;
    for (int n = 1; n <= 256; n++) {
        const int64_t zlen = (Z + n - 1) / n;
        if (n > 1 && zlen < 4 * R) break;
        const double wgs = (double)tiles * (double)((Z + zlen - 1) / zlen);
        const double cost = (wgs <= slots ? 1.0 : wgs / slots) * (double)(zlen + 2 * R);
        if (n == 1 || cost < best_cost) {
            best = n;
            best_cost = cost;
        }
    }
    return best;
}

/* Ring kernel launcher.  Returns false when the shape is outside it (32-bit buffer offsets: a chunk with its lead-in
 * planes must stay below 4 GiB), and the caller falls back to the first form. */
template <int R, int BR, bool HAS_OUT, bool HAS_DOG, int PF, int XO>
static bool launch_ring_t(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, const fb_taps2 &t)
{
    using C = fb_ring_cfg<R, BR, PF, XO>;
    static int resident = 0; /* workgroups of this instantiation one CU holds (LDS, registers) */
    if (resident == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blur_fused_ring_kernel<R, BR, HAS_OUT, HAS_DOG, PF, XO>, C::NT, 0) != hipSuccess || n < 1) n = 1;
        resident = n;
    }
    const int64_t plane_bytes = X * Y * 4;
    const int64_t max_planes = (int64_t)0xFFFFFFF0ll / plane_bytes - 2 * R - 2; /* planes per chunk the offsets can address */
    if (max_planes < 4 * R || max_planes < 1) return false;
    const int tiles_x = (int)((X + FB_TX - 1) / FB_TX), tiles_y = (int)((Y + C::TY - 1) / C::TY);
    const long long tiles = (long long)tiles_x * tiles_y;
    int n = fused_chunks(R, Z, tiles, resident);
    if ((Z + n - 1) / n > max_planes) n = (int)((Z + max_planes - 1) / max_planes);
    const int zlen = (int)((Z + n - 1) / n);
    const int nch = (int)((Z + zlen - 1) / zlen);
    const long long total = tiles * nch;
    const long long per = (total + 7) / 8;
    static const char *penv = getenv("SIFT3D_RING_PRIO"); /* A/B aid: 0 = no issue priority for the x-pass wavefronts */
    const int prio = penv ? atoi(penv) : 0;
    hipLaunchKernelGGL((blur_fused_ring_kernel<R, BR, HAS_OUT, HAS_DOG, PF, XO>), dim3((unsigned)(8 * per)), dim3(C::NT), 0, s, in, out, dog, (int)X,
                       (int)Y, (int)Z, zlen, tiles_x, tiles_y, total, prio, t);
    return true;
}

template <int R, int BR, int PF, int XO>
static bool launch_ring_xo(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, const fb_taps2 &t)
{
    if (out && dog) return launch_ring_t<R, BR, true, true, PF, XO>(s, in, out, dog, X, Y, Z, t);
    if (out) return launch_ring_t<R, BR, true, false, PF, XO>(s, in, out, dog, X, Y, Z, t);
    return launch_ring_t<R, BR, false, true, PF, XO>(s, in, out, dog, X, Y, Z, t);
}

/* SIFT3D_RING_XO (A/B aid): outputs per x-pass lane, 8 or 4 (4 only with one row per thread: it needs the wavefronts) */
template <int R, int BR, int PF>
static bool launch_ring_pf(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, const fb_taps2 &t)
{
    if constexpr (BR == 1 && PF == 1) {
        const char *env = getenv("SIFT3D_RING_XO");
        const int xo = env ? atoi(env) : 8;
        if (xo == 4) return launch_ring_xo<R, BR, PF, 4>(s, in, out, dog, X, Y, Z, t);
    }
    return launch_ring_xo<R, BR, PF, 8>(s, in, out, dog, X, Y, Z, t);
}

/* SIFT3D_RING_PF (A/B aid): planes of window prefetch, 1 or 2 */
template <int R, int BR>
static bool launch_ring_br(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, const fb_taps2 &t)
{
    const char *env = getenv("SIFT3D_RING_PF");
    /* by measurement at 512^3: two planes in flight and one workgroup per CU for the two-rows-per-thread mapping (7 to 13
     * taps); the 1024-thread mapping has 128 registers per thread and keeps one plane */
    const int pf = env ? atoi(env) : (BR == 2 ? 2 : 1);
    if (pf == 2) return launch_ring_pf<R, BR, 2>(s, in, out, dog, X, Y, Z, t);
    return launch_ring_pf<R, BR, 1>(s, in, out, dog, X, Y, Z, t);
}

/* SIFT3D_RING_BR (A/B aid): rows per thread, 1 or 2 */
template <int R>
static bool launch_ring(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, const fb_taps2 &t)
{
    const char *env = getenv("SIFT3D_RING_BR");
    /* by measurement: two rows per thread up to 13 taps and one row for 17 at 512^3 and 256^3; one row below 2^22 voxels,
     * where a volume has fewer tiles than the chip has CUs and sixteen wavefronts per workgroup help */
    const int br = env ? atoi(env) : ((R >= 7 || X * Y * Z < (1ll << 22)) ? 1 : 2);
    if (br == 1) return launch_ring_br<R, 1>(s, in, out, dog, X, Y, Z, t);
    return launch_ring_br<R, 2>(s, in, out, dog, X, Y, Z, t);
}

template <int R, int NBUF>
static void launch_fused_dma_n(hipStream_t s, const float *in, float *out, float *dog, const float *zeros, int64_t X, int64_t Y,
                               int64_t Z, const fb_taps2 &t)
{
    constexpr int TY = 32;
    static int resident = 0; /* workgroups of this instantiation one CU holds (LDS, registers) */
    if (resident == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blur_fused_dma_kernel<R, TY, NBUF>, 16 * TY, 0) != hipSuccess || n < 1) n = 1;
        resident = n;
    }
    const int tiles_x = (int)((X + FB_TX - 1) / FB_TX), tiles_y = (int)((Y + TY - 1) / TY);
    const long long tiles = (long long)tiles_x * tiles_y;
    const int n = fused_chunks(R, Z, tiles, resident);
    const int zlen = (int)((Z + n - 1) / n);
    const int nch = (int)((Z + zlen - 1) / zlen);
    const long long total = tiles * nch;
    const long long per = (total + 7) / 8;
    /* the second half of the zero page's 512 bytes is the write sink of lanes outside the volume */
    hipLaunchKernelGGL((blur_fused_dma_kernel<R, TY, NBUF>), dim3((unsigned)(8 * per)), dim3(16 * TY), 0, s, in, out, dog, zeros,
                       const_cast<float *>(zeros) + 64, (int)X, (int)Y, (int)Z, zlen, tiles_x, tiles_y, total, t);
}

template <int R>
static void launch_fused_dma(hipStream_t s, const float *in, float *out, float *dog, const float *zeros, int64_t X, int64_t Y,
                             int64_t Z, const fb_taps2 &t)
{
    const char *env = getenv("SIFT3D_FUSED_NBUF"); /* tuning / test aid: ring depth 3 or 4 */
    const int nbuf = env ? atoi(env) : (R == 5 ? 3 : 4); /* by measurement at 512^3 */
    if (nbuf == 3) launch_fused_dma_n<R, 3>(s, in, out, dog, zeros, X, Y, Z, t);
    else launch_fused_dma_n<R, 4>(s, in, out, dog, zeros, X, Y, Z, t);
}

template <int R, int TY>
static void launch_fused_ty(hipStream_t s, const float *in, float *out, float *dog, const float *zeros, int64_t X, int64_t Y,
                            int64_t Z, const fb_taps2 &t)
{
    static int resident = 0; /* workgroups of this instantiation one CU holds (registers) */
    if (resident == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blur_fused_kernel<R, TY, FB_PF(R)>, 16 * TY, 0) != hipSuccess || n < 1) n = 1;
        resident = n;
    }
    const int tiles_x = (int)((X + FB_TX - 1) / FB_TX), tiles_y = (int)((Y + TY - 1) / TY);
    const long long tiles = (long long)tiles_x * tiles_y;
    const int n = fused_chunks(R, Z, tiles, resident);
    const int zlen = (int)((Z + n - 1) / n);
    const int nch = (int)((Z + zlen - 1) / zlen);
    const long long total = tiles * nch;
    const long long per = (total + 7) / 8;
    hipLaunchKernelGGL((blur_fused_kernel<R, TY, FB_PF(R)>), dim3((unsigned)(8 * per)), dim3(16 * TY), 0, s, in, out, dog, zeros, (int)X,
                       (int)Y, (int)Z, zlen, tiles_x, tiles_y, total, t);
}

template <int R>
static void launch_fused(hipStream_t s, const float *in, float *out, float *dog, const float *zeros, int64_t X, int64_t Y,
                         int64_t Z, const fb_taps2 &t)
{
    {
        const char *v = getenv("SIFT3D_FUSED_V"); /* A/B aid: 1 = the first form of the march; default = the ring kernel */
        /* the ring kernel's z pass shares the product of taps j and 2R-j: only for taps that are symmetric bit for bit
         * (sift3d_gauss_taps' always are: (j-R)^2 and the normalising sum are the same for both) */
        bool sym = true;
        for (int j = 0; j < R; j++) sym = sym && __builtin_bit_cast(unsigned, t.f[j].x) == __builtin_bit_cast(unsigned, t.f[2 * R - j].x);
        if (sym && !(v && atoi(v) == 1) && launch_ring<R>(s, in, out, dog, X, Y, Z, t)) return;
    }
    if constexpr (R >= 5 && R <= 7) { /* 17 taps: arithmetic-bound, the register window is faster */
        const char *dma = getenv("SIFT3D_FUSED_DMA"); /* A/B aid: 0 = the register-window kernel */
        if (!dma || atoi(dma) != 0) return launch_fused_dma<R>(s, in, out, dog, zeros, X, Y, Z, t);
    }
    const char *env = getenv("SIFT3D_FUSED_TY"); /* tuning / test aid: 16 or 32 */
    /* 11 taps: the 512-thread workgroup would need 160 registers per thread at three wavefronts per SIMD, i.e. one
     * workgroup per CU; two 256-thread ones do better there */
    const int ty = env ? atoi(env) : ((Y >= 64 && R != 5) ? 32 : 16);
    if (ty == 32) launch_fused_ty<R, 32>(s, in, out, dog, zeros, X, Y, Z, t);
    else launch_fused_ty<R, 16>(s, in, out, dog, zeros, X, Y, Z, t);
}

/* Returns hipErrorNotSupported when the shape is outside this kernel (the caller then runs the three-pass
 * path): rows must be whole 16-byte vectors and the filter at most 17 taps.  out or dog may be NULL.
 * zeros: 512 bytes of device memory: the first 256 hold 0.0f and are only read, the second 256 are a write sink. */
hipError_t sift3d_launch_blur_fused(hipStream_t s, const float *in, float *out, float *dog, const float *zeros, int64_t X,
                                    int64_t Y, int64_t Z, const float *taps, int ntaps)
{
    const int R = ntaps / 2;
    if (R < 1 || R > SIFT3D_FAST_MAX_R || X % 4 != 0 || X * Y >= (1ll << 29) || (!out && !dog) || !zeros)
        return hipErrorNotSupported;
    fb_taps2 t;
    for (int i = 0; i < 2 * SIFT3D_FAST_MAX_R + 1; i++) t.f[i] = v2f(i < ntaps ? taps[i] : 0.0f);
    switch (R) {
    case 1: launch_fused<1>(s, in, out, dog, zeros, X, Y, Z, t); break;
    case 2: launch_fused<2>(s, in, out, dog, zeros, X, Y, Z, t); break;
    case 3: launch_fused<3>(s, in, out, dog, zeros, X, Y, Z, t); break;
    case 4: launch_fused<4>(s, in, out, dog, zeros, X, Y, Z, t); break;
    case 5: launch_fused<5>(s, in, out, dog, zeros, X, Y, Z, t); break;
    case 6: launch_fused<6>(s, in, out, dog, zeros, X, Y, Z, t); break;
    case 7: launch_fused<7>(s, in, out, dog, zeros, X, Y, Z, t); break;
    default: launch_fused<8>(s, in, out, dog, zeros, X, Y, Z, t); break;
    }
    return hipGetLastError();
}
