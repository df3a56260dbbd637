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
 * A workgroup (512 threads, or 1024) owns a 64 x 32 (x, y) tile and marches along z:
 *   A  x pass    a lane of the x-pass wavefronts keeps an aligned window of one
 *                input row in registers (loaded one or two planes ahead) and
 *                produces 8 outputs of that row; the 32 + 2R rows of the tile and
 *                its y halo go to one of two LDS buffers (one barrier per plane);
 *   B  y pass    every thread produces a 2 x 2 (or 2 x 1) block from LDS;
 *   C  z pass    the block feeds the thread's 2R+1 partial sums of the output
 *                planes it touches, all in registers (shift form: every slot a
 *                compile-time register); the finished plane is stored together
 *                with input - level, the input voxel (the "previous level" of the
 *                DoG) coming from an LDS ring the x pass filled R steps earlier.
 * The z range is cut into chunks that recompute 2R lead-in planes.  The register
 * file (512 KB per CU) holds what an LDS ring of 2R+1 planes (78 KB per tile for
 * 17 taps) would hold otherwise.
 *
 * Arithmetic contract: identical to kernels_volume.hip (and to the reference's
 * CPU path, R/src_common/GaussBlur3D.cpp:43-61,329-479): every pass is
 *   acc = 0; for j ascending: acc = acc + f[j]*v[j]
 * with separately rounded multiply and add (-ffp-contract=off; the packed
 * v_pk_mul_f32 / v_pk_add_f32 forms are the same IEEE operations on two lanes),
 * zeros outside the volume, float32 between the passes.  A zero tap adds +0 to an
 * accumulator that is never -0, so zero rows / planes and skipped taps agree.
 *
 * History: a first form of this march (round 1: window through per-vector pointers
 * and a zero page, DoG input re-read from memory, z pass behind a switch on the
 * plane's phase, an LDS-DMA input ring for 11 and 13 taps) lived here beside the
 * kernel below until round 3; DESIGN.md section 4 keeps its measurements.
 */
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
/* Cache policy of the level / DoG stores (the aux operand of buffer_store): 2 = nt, non-temporal.  Measured at 512^3 against
 * 0 (default), 3 (nt + sc0), 17 (sc0 + sc1) and 19 (nt + sc0 + sc1): launch times within 2 %, and the L2 read misses of the
 * two instantiations that re-fetch part of their tile halo (7 and 9 taps with DoG: 4.9 and 5.4 B/voxel through the fabric
 * against 4.2 - 4.4 for the others) stay where they are (4.8 - 5.3 / 5.2 - 6.0): the policy of the stores is not what
 * evicts those lines (profiles/r03_store_policy.txt). */
#define FB_STORE_AUX 2

struct fb_taps2 {
    v2f f[2 * SIFT3D_FAST_MAX_R + 1]; /* (f[j], f[j]): a 64-bit scalar operand of the packed multiply */
};

/* ------------------------------------------------------------------------------------------------------------------
 * The march ("ring" kernel, round 2; the second form of this kernel).  Against the first form: same tile, same three
 * passes, same arithmetic; what changed is everything around the arithmetic:
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

/* XO = outputs per lane of the x pass: 8, i.e. a wavefront filters 8 rows of the tile and the 32 + 2R rows are 5 or 6
 * wavefronts' worth.  (4 outputs per lane -- 10 to 12 wavefronts, three per SIMD instead of an uneven two / one -- was
 * built for the 1024-thread mapping in round 2: bit-identical, and no faster: 0.402 against 0.401 ms at 17 taps.) */
#define FB_XO 8
/* TXv x TYv: the (x, y) tile, 2 048 voxels either way.  64 x 32 is the shape of rounds 1 - 3; 128 x 16 (round 4) makes the row
 * segment a workgroup reads and writes 512 bytes instead of 256 -- the zero-arithmetic march streams 6 - 8 % better on such
 * tiles ("Tile shapes") -- and pays for it with a taller relative y halo: the x pass filters 16 + 2R rows
 * for 16 instead of 32 + 2R for 32. */
template <int R, int BR, int PF = 1, int TXv = FB_TX, int TYv = 32>
struct fb_ring_cfg {
    static constexpr int XO = FB_XO;
    static constexpr int TX = TXv;
    static constexpr int TY = TYv;
    static constexpr int CP = TX / 2;                           /* column pairs of the tile: the y / z pass's lanes along x */
    static constexpr int NT = 1024 / BR;
    static_assert(CP * (TY / BR) == NT, "every thread owns a 2 x BR block of the tile");
    static constexpr int NR = TY + 2 * R;
    static constexpr int LPR = TX / XO;                         /* lanes per row of the x pass */
    static constexpr int RPW = 64 / LPR;                        /* rows per x-pass wavefront */
    static constexpr int XW = (NR + RPW - 1) / RPW;             /* wavefronts with an x-pass role */
    static constexpr int P1ROWS = XW * RPW;                     /* rows those wavefronts write (>= NR; the surplus rows are never read) */
    static constexpr int S = R + 2;                /* DoG-input ring: written for plane z+1 while plane z-R is read */
    /* SINGLE: the configuration meant to run one workgroup per CU (two rows per thread, two or three planes of prefetch).  Its
     * LDS request is padded past half of the CU's 160 KiB so that a second workgroup can never become resident, also on
     * volumes with more tiles than CUs: two resident workgroups with two planes in flight each stream markedly slower
     * (0.38 against 0.315 ms per 7-tap launch at 512^3). */
    static constexpr bool SINGLE = PF >= 2 && BR == 2;
    static constexpr int LDS_NEEDED = 2 * P1ROWS * TX + S * TY * TX;
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
/* One workgroup's march: tile and z chunk number wi of the launch's list (tiles_x * tiles_y tiles by z chunks of zlen planes).
 * A function of its own so that a launch is a thin wrapper around it. */
template <int R, int BR, bool HAS_OUT, bool HAS_DOG, int PF, int TXv = FB_TX, int TYv = 32>
__device__ __forceinline__ void blur_fused_ring_body(float *__restrict__ lds, long long wi, const float *__restrict__ in, float *__restrict__ out,
                                                     float *__restrict__ dog, int X, int Y, int Z, int zo0, int zo1, int zlen, int tiles_x,
                                                     int tiles_y, const fb_taps2 &t)
{
    using C = fb_ring_cfg<R, BR, PF, TXv, TYv>;
    constexpr int U = 2 * R + 1, XO = C::XO;
    constexpr int TX = C::TX, CP = C::CP;
    constexpr int TY = C::TY, NR = C::NR, XW = C::XW, P1ROWS = C::P1ROWS, S = C::S;
    constexpr int H4 = ((R + 3) / 4) * 4; /* window halo, whole 16-byte vectors */
    constexpr int WIN = XO + 2 * H4, NV = WIN / 4;
    constexpr int LPR = C::LPR;
    static_assert(XO == 8, "a lane filters two 16-byte vectors of a row");
    constexpr int P1PL = P1ROWS * TX, PVPL = TY * TX;
    static_assert(XW * 64 <= C::NT, "the x-pass wavefronts are wavefronts of the workgroup");
    float *const P1b = lds;
    float *const pvb = lds + 2 * P1PL;
    const int tx = (int)(wi % tiles_x);
    const int ty = (int)((wi / tiles_x) % tiles_y);
    const int chunk = (int)(wi / ((long long)tiles_y * tiles_x));
    const int x0 = tx * TX, y0 = ty * TY;
    /* output planes [zo0, zo1) of the volume (the whole of it, or a window: a Z-slab rank filters its boundary bands
     * first); the input is read wherever the filter reaches, zeros outside [0, Z) */
    const int zc0 = zo0 + chunk * zlen;
    const int zc1 = zc0 + zlen < zo1 ? zc0 + zlen : zo1;
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
    const int bcp = tid % CP, brow = (tid / CP) * BR;
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
            for (int q = 0; q < XO / 4; q++) *reinterpret_cast<v4f *>(&pvs[(ar - R) * TX + axs + 4 * q]) = win[B][H4 / 4 + q];
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
            *reinterpret_cast<v4f *>(&P1[ar * TX + axs + 4 * q]) = r0;
        }
    };

    auto store_plane = [&](bool live, unsigned so, const v2f(&lv)[BR], const v2f(&dg)[BR]) {
        if constexpr (HAS_OUT) {
            const __amdgpu_buffer_rsrc_t r = store_rsrc(out_base, live);
#pragma unroll
            for (int q = 0; q < BR; q++) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, lv[q]), r, (int)soff[q], (int)so, FB_STORE_AUX);
        }
        if constexpr (HAS_DOG) {
            const __amdgpu_buffer_rsrc_t r = store_rsrc(dog_base, live);
#pragma unroll
            for (int q = 0; q < BR; q++) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, dg[q]), r, (int)soff[q], (int)so, FB_STORE_AUX);
        }
    };

    v2f acc[BR][U];
#pragma unroll
    for (int r = 0; r < BR; r++)
#pragma unroll
        for (int i = 0; i < U; i++) acc[r][i] = v2f(0.0f);
    int cur = 0;
    int wslot = 0; /* ring slot of the plane the x pass is working on; the plane stored this step sits in wslot + 1 (mod R+2) */

    /* (The x-pass wavefronts carry the longest step -- x + y + z pass against y + z -- and everyone meets them at the barrier;
     * giving them issue priority over the wavefronts that share their SIMD, s_setprio 2, changed nothing: 0.390 against
     * 0.387 - 0.395 ms at 17 taps.  They are not short of issue slots.) */
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1 % PF>;
    using B2 = std::integral_constant<int, 2 % PF>;
    if (xrole) { /* plane zfirst + k lives in window buffer k % PF */
        load_window(zfirst, B0{});
        x_pass(P1b, pvb, B0{});        /* plane zfirst (zeros when it lies before the volume) */
        if constexpr (PF == 3) {
            load_window(zfirst + 1, B1{});
            load_window(zfirst + 2, B2{});
            load_window(zfirst + 3, B0{});
        } else if constexpr (PF == 2) {
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
        for (int q = 0; q < U + BR - 1; q++) p[q] = *reinterpret_cast<const v2f *>(&P1[(brow + q) * TX + 2 * bcp]);
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
                const float *pvp = pvb + rslot * PVPL + brow * TX + 2 * bcp;
#pragma unroll
                for (int r = 0; r < BR; r++) dg[r] = *reinterpret_cast<const v2f *>(pvp + r * TX) - a[r];
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
    if constexpr (PF == 3) {
        int zin = zfirst;
        for (; zin + 2 <= zlast; zin += 3) {
            step(zin, B1{});
            step(zin + 1, B2{});
            step(zin + 2, B0{});
        }
        if (zin <= zlast) step(zin, B1{});
        if (zin + 1 <= zlast) step(zin + 1, B2{});
    } else if constexpr (PF == 2) {
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

template <int R, int BR, bool HAS_OUT, bool HAS_DOG, int PF, int TXv = FB_TX, int TYv = 32>
__global__ __launch_bounds__(1024 / BR, (fb_ring_cfg<R, BR, PF, TXv, TYv>::WAVES_PER_SIMD)) void blur_fused_ring_kernel(
    const float *__restrict__ in, float *__restrict__ out, float *__restrict__ dog, int X, int Y, int Z, int zo0, int zo1, int zlen,
    int tiles_x, int tiles_y, long long total, fb_taps2 t)
{
    using C = fb_ring_cfg<R, BR, PF, TXv, TYv>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    const long long lin = blockIdx.x;
    const long long per = (total + 7) / 8;
    const long long wi = (lin % 8) * per + lin / 8; /* XCD-aware tile order, as in the first form */
    if (wi >= total) return;
    blur_fused_ring_body<R, BR, HAS_OUT, HAS_DOG, PF, TXv, TYv>(lds, wi, in, out, dog, X, Y, Z, zo0, zo1, zlen, tiles_x, tiles_y, t);
}

/* chunks along z: enough workgroups to fill every CU's resident slots while the 2R lead-in planes stay cheap */
static int fused_chunks(int R, int64_t Z, long long tiles, int resident, int forced, double slots_given = 0.0)
{
    if (forced >= 1) return forced;
    /* time ~ rounds of resident workgroups x planes marched per workgroup */
    const double slots = slots_given > 0.0 ? slots_given : 256.0 * resident;
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

/* Returns false when the shape is outside the kernel (32-bit buffer offsets: a chunk with its lead-in planes must stay
 * below 4 GiB -- a volume whose planes are that large gets more z chunks, and only a plane pair beyond 4 GiB has none) */
template <int R, int BR, bool HAS_OUT, bool HAS_DOG, int PF, int TXv = FB_TX, int TYv = 32>
static bool launch_ring_t(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1,
                          const fb_taps2 &t, int forced_chunks)
{
    using C = fb_ring_cfg<R, BR, PF, TXv, TYv>;
    static int resident = 0; /* workgroups of this instantiation one CU holds (LDS, registers) */
    if (resident == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blur_fused_ring_kernel<R, BR, HAS_OUT, HAS_DOG, PF, TXv, TYv>, C::NT, 0) != hipSuccess || n < 1) n = 1;
        resident = n;
    }
    const int64_t plane_bytes = X * Y * 4;
    const int64_t max_planes = (int64_t)0xFFFFFFF0ll / plane_bytes - 2 * R - 2; /* planes per chunk the offsets can address */
    if (max_planes < 1) return false;
    const int tiles_x = (int)((X + C::TX - 1) / C::TX), tiles_y = (int)((Y + C::TY - 1) / C::TY);
    const long long tiles = (long long)tiles_x * tiles_y;
    const int64_t Zo = zo1 - zo0; /* planes to produce */
    int n = fused_chunks(R, Zo, tiles, resident, forced_chunks);
    if ((Zo + n - 1) / n > max_planes) n = (int)((Zo + max_planes - 1) / max_planes);
    const int zlen = (int)((Zo + n - 1) / n);
    const int nch = (int)((Zo + zlen - 1) / zlen);
    const long long total = tiles * nch;
    const long long per = (total + 7) / 8;
    hipLaunchKernelGGL((blur_fused_ring_kernel<R, BR, HAS_OUT, HAS_DOG, PF, TXv, TYv>), dim3((unsigned)(8 * per)), dim3(C::NT), 0, s, in, out, dog, (int)X,
                       (int)Y, (int)Z, (int)zo0, (int)zo1, zlen, tiles_x, tiles_y, total, t);
    return true;
}

template <int R, int BR, int PF, int TXv = FB_TX, int TYv = 32>
static bool launch_ring_pf(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1,
                           const fb_taps2 &t, int chunks)
{
    if (out && dog) return launch_ring_t<R, BR, true, true, PF, TXv, TYv>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks);
    if (out) return launch_ring_t<R, BR, true, false, PF, TXv, TYv>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks);
    return launch_ring_t<R, BR, false, true, PF, TXv, TYv>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks);
}

/* The two mappings, by measurement at 512^3 and 256^3: two rows per thread, two planes of window
 * prefetch and one workgroup per CU up to 13 taps; one row per thread (1024 threads, 128 registers) with one plane for 15
 * and 17 taps, and for every filter below 2^22 voxels, where a volume has fewer tiles than the chip has CUs and sixteen
 * wavefronts per workgroup help.  tune->rows_per_thread forces one of the two (tests run both on every shape). */
template <int R>
static bool launch_ring(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1,
                        const fb_taps2 &t, const sift3d_blur_tuning *tune)
{
    const int forced = tune ? tune->rows_per_thread : 0;
    const int br = forced == 1 || forced == 2 ? forced : ((R >= 7 || X * Y * (zo1 - zo0) < (1ll << 22)) ? 1 : 2);
    const int chunks = tune ? tune->z_chunks : 0;
    if (br == 1) return launch_ring_pf<R, 1, 1>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks);
    /* tile shape of the two-rows-per-thread mapping: tune->tile 0 = by measurement (below), 1 = 64 x 32, 2 = 128 x 16 */
    /* by measurement at 512^3 (profiles/r04_tile_ab.txt, dense random data, ms per launch 64 x 32 -> 128 x 16): 7 taps level only
     * 0.217 -> 0.201 - 0.207, 9 taps level + DoG 0.333 - 0.337 -> 0.318 - 0.323, 7 taps level + DoG 0.329 - 0.331 -> 0.321 - 0.326; no
     * gain at 11 taps (0.356 - 0.358 both) and a loss where the taller y halo meets more arithmetic or three planes of prefetch:
     * 13 taps 0.364 - 0.368 -> 0.370 - 0.375, 9 taps level only 0.216 - 0.220 -> 0.225 - 0.234 */
    const int tile = tune ? tune->tile : 0;
    const bool both = out && dog;
    const bool wide = tile == 2 || (tile == 0 && ((R == 3) || (R == 4 && both)));
    if constexpr (R <= 6)
        if (wide && X >= 128) {
            if constexpr (R <= 4)
                if (!(out && dog)) return launch_ring_pf<R, 2, 3, 128, 16>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks);
            return launch_ring_pf<R, 2, 2, 128, 16>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks);
        }
    /* three planes of window prefetch where the registers are there and only one array is stored (7 and 9 taps, level
     * only: 0.213 / 0.224 ms at 512^3 against 0.225 - 0.232 / 0.233 - 0.237 with two; with the DoG store beside it three planes
     * change nothing: 0.324 / 0.338 against 0.328 / 0.334 - 0.342; four planes, level only: 0.225 / 0.220, no better than three) */
    if constexpr (R <= 4)
        if (!(out && dog)) return launch_ring_pf<R, 2, 3>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks);
    return launch_ring_pf<R, 2, 2>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks);
}

/* ------------------------------------------------------------------------------------------------------------------
 * The coarse octaves' levels in ONE persistent launch (round 4, second form; the first -- three passes per level through
 * L2, ninety grid barriers -- lost: profiles/r04_coarse_chain.txt).  After the first two octaves a 512^3 pyramid is sixty
 * dependent launches of 7 - 25 us that queue, one by one, behind the grids of the first octave's extrema passes.  Here
 * every level of those octaves is the march above -- x, y, z and the DoG in one step, as virtual workgroups of a resident
 * grid -- so an octave is FOUR dependent steps and a subsample instead of thirteen launches, with a grid-wide barrier
 * where a launch boundary was; the octaves of at most SIFT3D_TINY_VOX voxels are built by workgroup 0 alone in LDS, as
 * tiny_octave_kernel does.  Being resident from its first instruction the launch does not queue behind anyone.
 * Only the levels: the extrema passes of these octaves stay launches of their own (they wait for this kernel, not for
 * each other).  Same arithmetic as everywhere (it IS the same code); D_0 and L_5 are not produced (lazy levels).
 * ------------------------------------------------------------------------------------------------------------------ */
#define FBC_THREADS 512
#ifdef SIFT3D_DEV /* development build: workgroup 0 leaves the 100 MHz clock around every grid barrier (tools/chain_phases.py) */
__device__ unsigned long long g_chain_clk[256];
__device__ unsigned g_chain_nclk;
#define FBC_CLK()                                                                                                        \
    do {                                                                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0 && g_chain_nclk < 256) g_chain_clk[g_chain_nclk++] = wall_clock64();     \
    } while (0)
extern "C" int sift3d_dev_chain_clocks(unsigned long long *out, int n)
{
    unsigned cnt = 0;
    if (hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_chain_nclk), sizeof cnt) != hipSuccess) return -1;
    if ((int)cnt > n) cnt = (unsigned)n;
    if (cnt && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_clk), sizeof(unsigned long long) * cnt) != hipSuccess) return -1;
    const unsigned zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_nclk), &zero, sizeof zero);
    return (int)cnt;
}
#else
#define FBC_CLK() do { } while (0)
#endif
constexpr int fbc_max(int a, int b) { return a > b ? a : b; }
/* the largest LDS request of the marches it runs (R = 3 .. 6, two rows per thread, two planes of prefetch, 64 x 32 tiles) */
constexpr int FBC_LDS_FLOATS = fbc_max(fbc_max(fb_ring_cfg<3, 2, 2>::LDS_FLOATS, fb_ring_cfg<4, 2, 2>::LDS_FLOATS),
                                       fbc_max(fb_ring_cfg<5, 2, 2>::LDS_FLOATS, fb_ring_cfg<6, 2, 2>::LDS_FLOATS));
static_assert(FBC_LDS_FLOATS >= 3 * SIFT3D_TINY_VOX + 2 * SIFT3D_FAST_MAX_R + 1, "the single-workgroup octaves fit the same LDS");
static_assert(fb_ring_cfg<6, 2, 2>::NT == FBC_THREADS, "the marches run with the launch's 512 threads");

__device__ __forceinline__ bool fbc_grid_barrier(unsigned *sync, unsigned G, unsigned &gen)
{
    /* arrival counter + generation word (zeroed by the launcher); release before arriving, acquire after leaving, at agent
     * scope: the levels one workgroup wrote are read by workgroups on other XCDs.  Every spin is bounded: a workgroup that
     * waits longer than any healthy run could need raises the abort word, everyone who sees it leaves, the host reports it. */
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
    return G > 1 ? __hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u : true;
}

/* 2 x 2 x 2 mean with the reference's association (FeatureIO.cpp:1532-1538, as subsample_kernel); dense rows */
__device__ __forceinline__ void fbc_subsample(unsigned wg, unsigned G, const float *__restrict__ in, int X, int Y, int Z, float *__restrict__ out)
{
    const unsigned ox = (unsigned)X / 2u, oy = (unsigned)Y / 2u, oz = (unsigned)Z / 2u, items = ox * oy * oz;
    const long long XY = (long long)X * Y;
    for (unsigned i = wg * FBC_THREADS + threadIdx.x; i < items; i += G * FBC_THREADS) {
        const unsigned x = i % ox, yz = i / ox, y = yz % oy, z = yz / oy;
        const float *p0 = in + (long long)(2 * z) * XY + (long long)(2 * y) * X + 2 * x;
        const float *p1 = p0 + XY;
        const float a00 = p0[0], a10 = p0[1], a01 = p0[X], a11 = p0[X + 1];
        const float b00 = p1[0], b10 = p1[1], b01 = p1[X], b11 = p1[X + 1];
        float s = 0.0f;
        s = s + (((a00 + a01) + a10) + a11);
        s = s + (((b00 + b01) + b10) + b11);
        out[i] = s * 0.125f;
    }
}

template <int AXIS>
__device__ __forceinline__ void fbc_lds_pass(const float *src, float *dst, int X, int Y, int Z, int N, const float *f, int nt)
{
    const int R = nt / 2;
    const int len = AXIS == 0 ? X : (AXIS == 1 ? Y : Z);
    const int st = AXIS == 0 ? 1 : (AXIS == 1 ? X : X * Y);
    for (int s = threadIdx.x; s < N; s += FBC_THREADS) {
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

/* an octave of at most SIFT3D_TINY_VOX voxels by ONE workgroup with the octave in LDS: all five levels and all five DoG
 * levels stored (the arithmetic and the stores of tiny_octave_kernel), then level 0 of the next octave */
__device__ __forceinline__ void fbc_tiny_octave(float *lds, const sift3d_blur_chain_octave &o, const sift3d_blur_chain_params &p)
{
    const int X = o.X, Y = o.Y, Z = o.Z, N = X * Y * Z;
    float *cur = lds, *a = lds + SIFT3D_TINY_VOX, *b = lds + 2 * SIFT3D_TINY_VOX, *taps = lds + 3 * SIFT3D_TINY_VOX;
    for (int s = threadIdx.x; s < N; s += FBC_THREADS) cur[s] = o.L[0][s];
    __syncthreads();
    for (int lvl = 0; lvl < 5; lvl++) {
        const int nt = p.ntaps[lvl];
        if ((int)threadIdx.x < nt) taps[threadIdx.x] = p.taps[lvl][threadIdx.x];
        __syncthreads();
        fbc_lds_pass<0>(cur, a, X, Y, Z, N, taps, nt);
        fbc_lds_pass<1>(a, b, X, Y, Z, N, taps, nt);
        fbc_lds_pass<2>(b, a, X, Y, Z, N, taps, nt);
        for (int s = threadIdx.x; s < N; s += FBC_THREADS) {
            const float v = a[s];
            if (lvl < 4) o.L[lvl + 1][s] = v;
            o.D[lvl][s] = cur[s] - v;
        }
        __syncthreads();
        if (lvl == 2 && o.next_L0) {
            fbc_subsample(0u, 1u, o.L[3], X, Y, Z, o.next_L0);
            __syncthreads();
        }
        float *tmp = cur; cur = a; a = tmp;
    }
}

template <int R, bool HAS_DOG>
__device__ __forceinline__ void fbc_level(float *lds, unsigned wg, unsigned G, const sift3d_blur_chain_octave &o, int j, const fb_taps2 &t)
{
    const long long total = (long long)o.tiles_x * o.tiles_y * o.nch[j - 1];
    for (long long wi = wg; wi < total; wi += G) {
        blur_fused_ring_body<R, 2, true, HAS_DOG, 2>(lds, wi, o.L[j - 1], o.L[j], HAS_DOG ? o.D[j - 1] : nullptr, o.X, o.Y, o.Z, 0, o.Z,
                                                     o.zlen[j - 1], o.tiles_x, o.tiles_y, t);
        __syncthreads(); /* the next tile starts with the LDS of this one */
    }
}

__global__ __launch_bounds__(FBC_THREADS, 2) void blur_chain_kernel(sift3d_blur_chain_params p)
{
    __shared__ __attribute__((aligned(16))) float lds[FBC_LDS_FLOATS];
    const unsigned wg = blockIdx.x, G = gridDim.x;
    unsigned gen = 0;
    FBC_CLK();
    for (int oi = 0; oi < p.n_grid; oi++) {
        const sift3d_blur_chain_octave &o = p.oct[oi];
        for (int j = 1; j <= 4; j++) {
            fb_taps2 t;
#pragma unroll
            for (int q = 0; q < 2 * SIFT3D_FAST_MAX_R + 1; q++) t.f[q] = v2f(q < p.ntaps[j - 1] ? p.taps[j - 1][q] : 0.0f);
            /* level 1's DoG (D_0) is never stored: the level below D_1 is read as L_0 - L_1 around the candidates */
            switch (p.ntaps[j - 1] / 2) {
            case 3: if (j == 1) fbc_level<3, false>(lds, wg, G, o, j, t); else fbc_level<3, true>(lds, wg, G, o, j, t); break;
            case 4: if (j == 1) fbc_level<4, false>(lds, wg, G, o, j, t); else fbc_level<4, true>(lds, wg, G, o, j, t); break;
            case 5: if (j == 1) fbc_level<5, false>(lds, wg, G, o, j, t); else fbc_level<5, true>(lds, wg, G, o, j, t); break;
            default: if (j == 1) fbc_level<6, false>(lds, wg, G, o, j, t); else fbc_level<6, true>(lds, wg, G, o, j, t); break;
            }
            /* level 0 of the next octave rides with level 4: both only read L_3 */
            if (j == 4 && o.next_L0) fbc_subsample(wg, G, o.L[3], o.X, o.Y, o.Z, o.next_L0);
            FBC_CLK();
            if (!fbc_grid_barrier(p.sync, G, gen)) return;
            FBC_CLK();
        }
    }
    if (wg != 0) return;
    for (int oi = p.n_grid; oi < p.n_oct; oi++) {
        fbc_tiny_octave(lds, p.oct[oi], p);
        FBC_CLK();
    }
}

hipError_t sift3d_launch_blur_chain(hipStream_t s, sift3d_blur_chain_params &p, int workgroups)
{
    if (p.n_oct < 1 || p.n_oct > SIFT3D_CHAIN_MAX_OCT || p.n_grid < 0 || p.n_grid > p.n_oct || workgroups < 1) return hipErrorInvalidValue;
    for (int j = 0; j < 4; j++)
        if (p.ntaps[j] / 2 < 3 || p.ntaps[j] / 2 > 6 || !(p.ntaps[j] & 1)) return hipErrorNotSupported;
    if (p.ntaps[4] < 3 || p.ntaps[4] > 2 * SIFT3D_FAST_MAX_R + 1) return hipErrorNotSupported;
    for (int j = 0; j < 5; j++)
        for (int q = 0; q < p.ntaps[j] / 2; q++)
            if (__builtin_bit_cast(unsigned, p.taps[j][q]) != __builtin_bit_cast(unsigned, p.taps[j][p.ntaps[j] - 1 - q])) return hipErrorNotSupported;
    for (int i = 0; i < p.n_oct; i++) {
        sift3d_blur_chain_octave &o = p.oct[i];
        if (o.X % 4 != 0 || (long long)o.X * o.Y >= (1ll << 29)) return hipErrorNotSupported;
        if (i >= p.n_grid) {
            if ((long long)o.X * o.Y * o.Z > SIFT3D_TINY_VOX) return hipErrorNotSupported;
            continue;
        }
        using C = fb_ring_cfg<6, 2, 2>;
        o.tiles_x = (o.X + C::TX - 1) / C::TX;
        o.tiles_y = (o.Y + C::TY - 1) / C::TY;
        for (int j = 0; j < 4; j++) { /* z chunks: enough virtual workgroups for the resident ones, lead-in planes kept cheap */
            const int R = p.ntaps[j] / 2;
            const int n = fused_chunks(R, o.Z, (long long)o.tiles_x * o.tiles_y, 1, 0, (double)workgroups);
            o.zlen[j] = (o.Z + n - 1) / n;
            o.nch[j] = (o.Z + o.zlen[j] - 1) / o.zlen[j];
        }
    }
    hipError_t e = hipMemsetAsync(p.sync, 0, sizeof(unsigned) * 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(blur_chain_kernel, dim3((unsigned)(p.n_grid > 0 ? workgroups : 1)), dim3(FBC_THREADS), 0, s, p);
    return hipGetLastError();
}

/* Returns hipErrorNotSupported when the shape is outside this kernel (the caller then runs the three-pass path): rows
 * must be whole 16-byte vectors, the filter at most 17 taps and symmetric bit for bit -- the z pass shares the product of
 * taps j and 2R-j (sift3d_gauss_taps' always are: (j-R)^2 and the normalising sum are the same for both).  out or dog may
 * be NULL.  [zo0, zo1): the planes to produce (zo1 < 0: all of them); planes outside the window are not written, the input
 * is read as far as the filter reaches. */
hipError_t sift3d_launch_blur_fused(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z,
                                    const float *taps, int ntaps, const sift3d_blur_tuning *tune, int64_t zo0, int64_t zo1)
{
    const int R = ntaps / 2;
    if (zo1 < 0) zo1 = Z; /* the default window: the whole volume */
    if (R < 1 || R > SIFT3D_FAST_MAX_R || ntaps != 2 * R + 1 || X % 4 != 0 || X * Y >= (1ll << 29) || (!out && !dog)) return hipErrorNotSupported;
    if (zo0 < 0 || zo1 > Z || zo1 <= zo0) return hipErrorInvalidValue;
    for (int j = 0; j < R; j++)
        if (__builtin_bit_cast(unsigned, taps[j]) != __builtin_bit_cast(unsigned, taps[2 * R - j])) return hipErrorNotSupported;
    fb_taps2 t;
    for (int i = 0; i < 2 * SIFT3D_FAST_MAX_R + 1; i++) t.f[i] = v2f(i < ntaps ? taps[i] : 0.0f);
    bool ok = false;
    switch (R) {
    case 1: ok = launch_ring<1>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune); break;
    case 2: ok = launch_ring<2>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune); break;
    case 3: ok = launch_ring<3>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune); break;
    case 4: ok = launch_ring<4>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune); break;
    case 5: ok = launch_ring<5>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune); break;
    case 6: ok = launch_ring<6>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune); break;
    case 7: ok = launch_ring<7>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune); break;
    default: ok = launch_ring<8>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune); break;
    }
    if (!ok) return hipErrorNotSupported;
    return hipGetLastError();
}
