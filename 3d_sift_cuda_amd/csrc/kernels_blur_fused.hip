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
/* HAS_SUB (round 4): the launch that makes level 3 also writes level 0 of the NEXT octave -- the 2 x 2 x 2 mean of the level
 * (fioSubSampleInterpolate, R/src_common/FeatureIO.cpp:1474-1554, its association kept: ((a00 + a01) + a10) + a11 per plane, the
 * even plane's sum first) -- from the finished planes it holds in registers: a thread's 2 x 2 block of a plane IS one such
 * block, and the block of the plane before waits in one register.  The subsample launch, which read the whole level again
 * (0.10 ms at 512^3), is gone; the z chunks start on even planes so that a pair never straddles two workgroups. */
/* STG (round 6): the second-dispatched half of the wavefronts runs half a step behind the first.  Every wavefront runs the same
 * program with one barrier per plane, so without it the two wavefronts of a SIMD (w and w + 4 of a 512-thread workgroup share one)
 * reach their LDS read burst, their arithmetic and the barrier together: the SIMD idles while both wait for the y pass's rows and
 * both want its issue slots afterwards (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  With STG, wavefronts NT/128 .. keep
 * the y-pass result g of a plane (and the DoG's input voxel) in registers across the barrier and run that plane's z pass, DoG
 * and stores at the START of the next step -- pure arithmetic, nothing to wait for -- while their SIMD partners wait for LDS;
 * the same operations on the same operands in the same order per accumulator, so the same bits.  Each (half, role) pair runs its
 * own copy of the march (wave-uniform branches in front of it), so every copy is straight-line code with exactly counted vmcnt
 * waits (and 170 registers at 13 taps instead of 222).  Measured (profiles/r06_stagger_ab.txt): the copies ALONE are 10 % slower
 * at 11 - 13 taps (exact waits let a wavefront start its x pass the moment its window is in, and both wavefronts of a SIMD then
 * collide on the y pass's LDS burst harder than before); with the stagger on top the 11- and 13-tap launches come out 2 - 3 %
 * ahead of the round-5 kernel, the 9-tap + DoG launch equal, the level-only launches 2 - 3 % behind -- so the launcher turns it
 * on from 11 taps up.  Under rocprofv3's counter runs, where the chip clocks differently, the same build is 12 - 14 % ahead at
 * 11 - 13 taps; un-profiled it is not.  Also measured and not kept: the stagger with the role tested at run time (3 % slower than
 * the round-5 kernel at 13 taps), three planes of prefetch under the stagger (no better than two), one plane (20 % slower). */
template <int R, int BR, bool HAS_OUT, bool HAS_DOG, int PF, int TXv = FB_TX, int TYv = 32, bool HAS_SUB = false, bool STG = false>
__global__ __launch_bounds__(1024 / BR, (fb_ring_cfg<R, BR, PF, TXv, TYv>::WAVES_PER_SIMD)) void blur_fused_ring_kernel(
    const float *__restrict__ in, float *__restrict__ out, float *__restrict__ dog, int X, int Y, int Z, int zo0, int zo1, int zlen,
    int tiles_x, int tiles_y, long long total, int order, fb_taps2 t, float *__restrict__ sub = nullptr)
{
    static_assert(!HAS_SUB || (BR == 2 && HAS_OUT), "the subsample rides on the 2 x 2 blocks of the two-rows-per-thread mapping");
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
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    float *const P1b = lds;
    float *const pvb = lds + 2 * P1PL;

    /* Which tile this workgroup takes.  Workgroup b of a launch runs on XCD b mod 8 (each with an L2 of its own), so the order
     * decides which tiles share an L2 and which addresses an XCD asks the fabric for.  wi = x + tiles_x * (y + tiles_y * chunk).
     *   order 1  XCD x walks the x-th eighth of the tiles in (x, y, chunk) order: whole rows of tiles per XCD (rounds 1 - 4);
     *   order 2  workgroup b takes tile b;
     *   order 3  column strips: XCD x owns column x mod tiles_x of the tiles (all of it, or -- fewer columns than XCDs -- a
     *            contiguous part of its (y, chunk) list), walking down y: a tile's y neighbours, with which it shares 2R of
     *            its TY + 2R input rows, are on its own XCD, and an XCD always asks for the same byte columns of every row.
     *            The launcher only passes 3 where the counts divide. */
    const long long lin = blockIdx.x;
    const long long per = (total + 7) / 8;
    long long wi;
    if (order == 3) {
        const int xcd = (int)(lin % 8);
        const long long j = lin / 8, M = total / tiles_x; /* (y, chunk) pairs of a column */
        if (tiles_x >= 8) {
            const int cols = tiles_x / 8;
            wi = (xcd + 8 * (j % cols)) + (long long)tiles_x * (j / cols);
        } else {
            const int g = 8 / tiles_x;
            wi = (xcd % tiles_x) + (long long)tiles_x * ((xcd / tiles_x) * (M / g) + j);
        }
    } else if (order == 2) {
        wi = lin;
    } else {
        wi = (lin % 8) * per + lin / 8;
    }
    if (wi >= total) return;
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
    /* HAS_SUB: this thread's voxel of the half-size volume (X / 2 by Y / 2 by Z / 2, dense rows); an odd last row / plane has no
     * partner and is dropped, as the reference drops it */
    const int oX = X >> 1, oY = Y >> 1, oZ = Z >> 1;
    unsigned suboff = FB_OOB;
    if constexpr (HAS_SUB)
        if (bx + 1 < X && y0 + brow + 1 < Y) suboff = (unsigned)(((y0 + brow) >> 1) * oX + (bx >> 1)) * 4u;
    const unsigned sub_plane_bytes = (unsigned)(oX * oY) * 4u;
    float subacc = 0.0f; /* ((a00 + a01) + a10) + a11 of the even plane of the pair */

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
    /* XC: 0 = the x-pass role is tested at run time (the form of rounds 2 - 5), 1 / 2 = this copy of the code is run by wavefronts
     * that have it / do not have it.  With the role known at compile time a step is straight-line code on every path and the
     * compiler's wait in front of the x pass is exact: vmcnt(19) .. (14) for the six vectors of a 13-tap window loaded two steps
     * earlier (two steps of 6 loads + 4 stores lie behind it); with the run-time test it has to assume that a step in between
     * may have skipped its loads and waits with vmcnt(13) .. (8), i.e. also for the first loads of the NEXT plane's window. */
    auto dropped_stores = [&]() { /* one step's stores through descriptors of no records: they only count in vmcnt */
        v2f z2[BR];
#pragma unroll
        for (int r = 0; r < BR; r++) z2[r] = v2f(0.0f);
        store_plane(false, 0u, z2, z2);
        if constexpr (HAS_SUB) {
            const __amdgpu_buffer_rsrc_t rsub = __builtin_amdgcn_make_buffer_rsrc((void *)sub, 0, 0, FB_RSRC_FLAGS);
            __builtin_amdgcn_raw_buffer_store_b32(0, rsub, (int)suboff, 0, 0);
        }
    };
    /* The windows of the first PF planes are requested with the loop's own pattern of stores between them, so that the
     * compiler's count of the operations behind a window at the loop's entry is the steady state's (it waits for the smaller of
     * the two): a step of the first half is "loads, stores", of the staggered half "stores, loads". */
    auto prologue = [&](auto late_c, auto xc) {
        constexpr bool LATE = decltype(late_c)::value;
        constexpr int XC = decltype(xc)::value;
        const bool xr = XC == 0 ? xrole : XC == 1;
        if (xr) { /* plane zfirst + k lives in window buffer k % PF */
            load_window(zfirst, B0{});
            x_pass(P1b, pvb, B0{});        /* plane zfirst (zeros when it lies before the volume) */
            if constexpr (XC == 0) { /* the form of rounds 2 - 5 */
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
            } else {
                if constexpr (PF == 3) {
                    load_window(zfirst + 1, B1{});
                    dropped_stores();
                    load_window(zfirst + 2, B2{});
                    dropped_stores();
                    load_window(zfirst + 3, B0{});
                } else if constexpr (PF == 2) {
                    load_window(zfirst + 1, B1{});
                    dropped_stores();
                    load_window(zfirst + 2, B0{});
                } else {
                    load_window(zfirst + 1, B0{});
                }
            }
        }
        if constexpr (!LATE || XC == 0) dropped_stores(); /* same vmcnt state as the loop's back edge */
        wslot = 1;
        lds_barrier();
    };

    /* z pass of the plane whose y-pass result is g, shift form, and the stores of the output plane it completes (zp - R).
     * Slot i holds the output plane that completes in i + 1 steps and therefore takes tap U-1-i of the new plane; the sum moves
     * one slot down as it is updated (a three-operand add reads slot i+1 and writes slot i, so the shift costs nothing), slot
     * U-1 restarts from 0 + f[0]*g.  Every slot index is a compile-time register without a switch on the plane's phase: the
     * first form's switch cost 2R+1 register copies per step (the phi nodes of its cases), 2R+1 copies of the z-pass code and a
     * basic-block boundary between the passes.  The taps are bit-symmetric (checked by the launcher), so f[j]*g and
     * f[U-1-j]*g are one product: R+1 multiplies instead of 2R+1, the additions and their order unchanged.
     * pv: the DoG's input voxels of plane zp - R (from the LDS ring). */
    auto z_tail = [&](int zp, const v2f(&g)[BR], const v2f(&pv)[BR]) {
        const bool emit = zp - R >= zc0; /* wave-uniform: plane zp - R is complete (it is < zc1 by construction of zlast) */
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
        const unsigned so = emit ? (unsigned)(zp - R - zc0) * plane_bytes : 0u;
        v2f dg[BR];
#pragma unroll
        for (int r = 0; r < BR; r++) dg[r] = HAS_DOG ? pv[r] - a[r] : v2f(0.0f);
        store_plane(emit, so, a, dg);
        if constexpr (HAS_SUB) {
            const int zo = zp - R; /* the plane just finished (meaningful when emit) */
            const float pa = ((a[0].x + a[1].x) + a[0].y) + a[1].y;
            float sm = 0.0f;
            sm = sm + subacc;
            sm = sm + pa;
            const bool pair_done = emit && (zo & 1) && (zo >> 1) < oZ; /* wave-uniform */
            /* issued on every step, like the other stores: through a descriptor of no records unless a pair is complete */
            const __amdgpu_buffer_rsrc_t rsub =
                __builtin_amdgcn_make_buffer_rsrc((void *)sub, 0, pair_done ? (int)((unsigned)oZ * sub_plane_bytes) : 0, FB_RSRC_FLAGS);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sm * 0.125f), rsub, (int)suboff,
                                                  pair_done ? (int)((unsigned)(zo >> 1) * sub_plane_bytes) : 0, 0);
            subacc = pa;
        }
    };
    /* LATE wavefronts: y-pass result and DoG input of the plane whose z pass is still to run */
    v2f gprev[BR], pvprev[BR];
#pragma unroll
    for (int r = 0; r < BR; r++) gprev[r] = pvprev[r] = v2f(0.0f);

    /* one step: y pass of plane zin from P1[cur], x pass of plane zin + 1 into the other buffer, z pass; plane zin - R is
     * complete and is stored (LATE: plane zin - 1 - R, at the start of the step; a first step with nothing pending shifts
     * zeros through accumulators that are zero, and its stores are dropped like every lead-in step's) */
    auto step = [&](int zin, auto bsel, auto late_c, auto xc) { /* bsel: the window buffer that holds plane zin + 1 */
        constexpr bool LATE = decltype(late_c)::value;
        constexpr int XC = decltype(xc)::value;
        const bool xr = XC == 0 ? xrole : XC == 1;
        const float *P1 = P1b + cur * P1PL;
        v2f p[U + BR - 1];
#pragma unroll
        for (int q = 0; q < U + BR - 1; q++) p[q] = *reinterpret_cast<const v2f *>(&P1[(brow + q) * TX + 2 * bcp]);
        if constexpr (LATE) {
            /* the y pass's rows are requested, and while they travel (and the SIMD partner, which asked for its rows at the same
             * moment, can only wait) the pending plane's z pass, DoG and stores run: nothing in them waits for anything.  The
             * fences keep the instruction scheduler from moving that block behind the y pass again (it did). */
            __builtin_amdgcn_sched_barrier(0);
            z_tail(zin - 1, gprev, pvprev);
            __builtin_amdgcn_sched_barrier(0);
        }
        v2f g[BR];
#pragma unroll
        for (int r = 0; r < BR; r++) {
            v2f a = v2f(0.0f);
#pragma unroll
            for (int j = 0; j < U; j++) a = a + t.f[j] * p[j + r];
            g[r] = a;
        }
        if (xr) {
            x_pass(P1b + (cur ^ 1) * P1PL, pvb + wslot * PVPL, bsel);
            load_window(zin + 1 + PF, bsel); /* the buffer is free again */
        }
        v2f pv[BR];
        if constexpr (HAS_DOG) {
            const int rslot = wslot + 1 == S ? 0 : wslot + 1;
            const float *pvp = pvb + rslot * PVPL + brow * TX + 2 * bcp;
#pragma unroll
            for (int r = 0; r < BR; r++) pv[r] = *reinterpret_cast<const v2f *>(pvp + r * TX);
        } else {
#pragma unroll
            for (int r = 0; r < BR; r++) pv[r] = v2f(0.0f);
        }
        if constexpr (LATE) {
#pragma unroll
            for (int r = 0; r < BR; r++) {
                gprev[r] = g[r];
                pvprev[r] = pv[r];
            }
        } else {
            z_tail(zin, g, pv);
        }
        cur ^= 1;
        wslot = wslot + 1 == S ? 0 : wslot + 1;
        lds_barrier(); /* the other P1 buffer and the ring slot are complete, every wavefront has read this P1 buffer */
    };
    /* the first 2R steps are lead-in: their stores are dropped */
    auto march = [&](auto late_c, auto xc) {
        constexpr bool LATE = decltype(late_c)::value;
        prologue(late_c, xc);
        if constexpr (PF == 3) {
            int zin = zfirst;
            for (; zin + 2 <= zlast; zin += 3) {
                step(zin, B1{}, late_c, xc);
                step(zin + 1, B2{}, late_c, xc);
                step(zin + 2, B0{}, late_c, xc);
            }
            if (zin <= zlast) step(zin, B1{}, late_c, xc);
            if (zin + 1 <= zlast) step(zin + 1, B2{}, late_c, xc);
        } else if constexpr (PF == 2) {
            int zin = zfirst;
            for (; zin + 1 <= zlast; zin += 2) {
                step(zin, B1{}, late_c, xc);     /* plane zfirst + 1 went to buffer 1 */
                step(zin + 1, B0{}, late_c, xc);
            }
            if (zin <= zlast) step(zin, B1{}, late_c, xc);
        } else {
            for (int zin = zfirst; zin <= zlast; zin++) step(zin, B0{}, late_c, xc);
        }
        if constexpr (LATE) z_tail(zlast, gprev, pvprev); /* the last plane's z pass and stores */
    };
    using XRT = std::integral_constant<int, 0>; /* role tested at run time */
    using XRY = std::integral_constant<int, 1>;
    using XRN = std::integral_constant<int, 2>;
    if constexpr (!STG) {
        march(std::false_type{}, XRT{});
    } else { /* wave-uniform branches: one copy of the march per (half, role) */
        if (wv < C::NT / 128) { /* the first-dispatched half */
            if (xrole) march(std::false_type{}, XRY{});
            else march(std::false_type{}, XRN{});
        } else {
            if (xrole) march(std::true_type{}, XRY{});
            else march(std::true_type{}, XRN{});
        }
    }
}

/* chunks along z: enough workgroups to fill every CU's resident slots while the 2R lead-in planes stay cheap */
static int fused_chunks(int R, int64_t Z, long long tiles, int resident, int forced)
{
    if (forced >= 1) return forced;
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

/* Returns false when the shape is outside the kernel (32-bit buffer offsets: a chunk with its lead-in planes must stay
 * below 4 GiB -- a volume whose planes are that large gets more z chunks, and only a plane pair beyond 4 GiB has none) */
template <int R, int BR, bool HAS_OUT, bool HAS_DOG, int PF, int TXv = FB_TX, int TYv = 32, bool HAS_SUB = false, bool STG = false>
static bool launch_ring_t(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1,
                          const fb_taps2 &t, int forced_chunks, float *sub = nullptr, int order = 0)
{
    using C = fb_ring_cfg<R, BR, PF, TXv, TYv>;
    static int resident = 0; /* workgroups of this instantiation one CU holds (LDS, registers) */
    if (resident == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blur_fused_ring_kernel<R, BR, HAS_OUT, HAS_DOG, PF, TXv, TYv, HAS_SUB, STG>, C::NT, 0) != hipSuccess || n < 1) n = 1;
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
    int zlen = (int)((Zo + n - 1) / n);
    if (HAS_SUB && (zlen & 1)) zlen++; /* a pair of planes never straddles two chunks (the window starts at plane 0) */
    const int nch = (int)((Zo + zlen - 1) / zlen);
    const long long total = tiles * nch;
    const long long per = (total + 7) / 8;
    /* the column-strip order needs counts that divide: 8 | tiles_x, or tiles_x | 8 with the (y, chunk) list of a column cut
     * evenly over the 8 / tiles_x XCDs that share it; everything else keeps the order of rounds 1 - 4 */
    if (order == 0) order = 1; /* by measurement: see DESIGN.md section 4 (round 5) */
    if (order == 3) {
        const long long M = (long long)tiles_y * nch;
        const bool ok = tiles_x >= 8 ? tiles_x % 8 == 0 : (8 % tiles_x == 0 && M % (8 / tiles_x) == 0);
        if (!ok) order = 1;
    }
    hipLaunchKernelGGL((blur_fused_ring_kernel<R, BR, HAS_OUT, HAS_DOG, PF, TXv, TYv, HAS_SUB, STG>), dim3((unsigned)(8 * per)), dim3(C::NT), 0, s, in, out, dog, (int)X,
                       (int)Y, (int)Z, (int)zo0, (int)zo1, zlen, tiles_x, tiles_y, total, order, t, sub);
    return true;
}

template <int R, int BR, int PF, int TXv = FB_TX, int TYv = 32, bool STG = false>
static bool launch_ring_pf(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1,
                           const fb_taps2 &t, int chunks, int order)
{
    if (out && dog) return launch_ring_t<R, BR, true, true, PF, TXv, TYv, false, STG>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks, nullptr, order);
    if (out) return launch_ring_t<R, BR, true, false, PF, TXv, TYv, false, STG>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks, nullptr, order);
    return launch_ring_t<R, BR, false, true, PF, TXv, TYv, false, STG>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks, nullptr, order);
}

/* The two mappings, by measurement at 512^3 and 256^3: two rows per thread, two planes of window
 * prefetch and one workgroup per CU up to 13 taps; one row per thread (1024 threads, 128 registers) with one plane for 15
 * and 17 taps, and for every filter below 2^22 voxels, where a volume has fewer tiles than the chip has CUs and sixteen
 * wavefronts per workgroup help.  tune->rows_per_thread forces one of the two (tests run both on every shape). */
template <int R>
static bool launch_ring(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1,
                        const fb_taps2 &t, const sift3d_blur_tuning *tune, float *sub, int *sub_done)
{
    const int forced = tune ? tune->rows_per_thread : 0;
    const int br = forced == 1 || forced == 2 ? forced : ((R >= 7 || X * Y * (zo1 - zo0) < (1ll << 22)) ? 1 : 2);
    const int chunks = tune ? tune->z_chunks : 0;
    const int order = tune ? tune->order : 0;
    /* the half-size volume beside the level (HAS_SUB): built for the one filter the pyramid asks it of -- level 3 is 11 taps
     * in every octave (oracle: sigma_extra[3]) -- with both arrays stored, the whole volume produced and rows that halve into
     * whole 16-byte vectors; anything else leaves *sub_done 0 and the caller launches the subsample itself */
    /* the half-step stagger of the second half of the wavefronts with one copy of the march per (half, role) -- the kernel's STG:
     * tune->stagger 0 = by measurement, 1 = off (the kernel of rounds 2 - 5), 2 = on; built for the two-rows-per-thread mapping
     * (eight wavefronts, two per SIMD) and the filters the pyramid launches (7 - 13 taps).  By measurement at 512^3
     * (profiles/r06_stagger_ab.txt; ms per launch off -> on): 11 taps + DoG + half-size volume 0.357 - 0.363 -> 0.346 - 0.354, 13 taps
     * + DoG 0.365 - 0.373 -> 0.352 - 0.367, 9 taps + DoG equal (0.300 - 0.307 / 0.297 - 0.305), the two level-only launches 2 - 3 %
     * SLOWER (7 taps 0.201 - 0.203 -> 0.205 - 0.212, 9 taps 0.217 - 0.225 -> 0.220 - 0.228): on from 11 taps up. */
    const int stg_knob = tune ? tune->stagger : 0;
    const bool stg = R >= 3 && R <= 6 && (stg_knob == 2 || (stg_knob == 0 && R >= 5));
    if constexpr (R == 5)
        if (sub && br == 2 && out && dog && zo0 == 0 && zo1 == Z && X % 8 == 0 && Z >= 2 && Y >= 2) {
            /* 128 x 16 under the stagger (round 6, profiles/r06_stagger_ab.txt section 8: 0.354 -> 0.340 - 0.344 ms at 512^3; without the
             * stagger the two tiles measured equal in round 4); SIFT3D_TUNE_FUSED_TILE forces either */
            const int tile5 = tune ? tune->tile : 0;
            const bool wide5 = (tile5 == 2 || (tile5 == 0 && stg)) && X >= 128;
#define FB_SUB(TXv, TYv)                                                                                                         \
    (stg ? launch_ring_t<5, 2, true, true, 2, TXv, TYv, true, true>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks, sub, order)        \
         : launch_ring_t<5, 2, true, true, 2, TXv, TYv, true, false>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks, sub, order))
            const bool ok = wide5 ? FB_SUB(128, 16) : FB_SUB(FB_TX, 32);
#undef FB_SUB
            if (ok && sub_done) *sub_done = 1;
            return ok;
        }
    if (br == 1) return launch_ring_pf<R, 1, 1>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks, order);
    /* tile shape of the two-rows-per-thread mapping: tune->tile 0 = by measurement (below), 1 = 64 x 32, 2 = 128 x 16 */
    /* by measurement at 512^3 (profiles/r04_tile_ab.txt, dense random data, ms per launch 64 x 32 -> 128 x 16): 7 taps level only
     * 0.217 -> 0.201 - 0.207, 9 taps level + DoG 0.333 - 0.337 -> 0.318 - 0.323, 7 taps level + DoG 0.329 - 0.331 -> 0.321 - 0.326; no
     * gain at 11 taps (0.356 - 0.358 both) and a loss where the taller y halo meets more arithmetic or three planes of prefetch:
     * 13 taps 0.364 - 0.368 -> 0.370 - 0.375, 9 taps level only 0.216 - 0.220 -> 0.225 - 0.234 */
    const int tile = tune ? tune->tile : 0;
    const bool both = out && dog;
    const bool wide = tile == 2 || (tile == 0 && ((R == 3) || (R == 4 && both)));
#define FB_PF(PFv, TXv, TYv)                                                                                                     \
    (stg ? launch_ring_pf<R, 2, PFv, TXv, TYv, (R >= 3 && R <= 6)>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks, order)           \
         : launch_ring_pf<R, 2, PFv, TXv, TYv, false>(s, in, out, dog, X, Y, Z, zo0, zo1, t, chunks, order))
    if constexpr (R <= 6)
        if (wide && X >= 128) {
            if constexpr (R <= 4)
                if (!(out && dog)) return FB_PF(3, 128, 16);
            return FB_PF(2, 128, 16);
        }
    /* three planes of window prefetch where the registers are there and only one array is stored (7 and 9 taps, level
     * only: 0.213 / 0.224 ms at 512^3 against 0.225 - 0.232 / 0.233 - 0.237 with two; with the DoG store beside it three planes
     * change nothing: 0.324 / 0.338 against 0.328 / 0.334 - 0.342; four planes, level only: 0.225 / 0.220, no better than three;
     * round 6, under the stagger, whose role copies leave the registers for it: 11 / 13 taps with three planes 0.351 - 0.354 /
     * 0.362 - 0.369 against 0.346 - 0.351 / 0.352 - 0.359 with two; ONE plane: 0.43 / 0.44) */
    if constexpr (R <= 4)
        if (!(out && dog)) return FB_PF(3, FB_TX, 32);
    return FB_PF(2, FB_TX, 32);
#undef FB_PF
}

/* Returns hipErrorNotSupported when the shape is outside this kernel (the caller then runs the three-pass path): rows
 * must be whole 16-byte vectors, the filter at most 17 taps and symmetric bit for bit -- the z pass shares the product of
 * taps j and 2R-j (sift3d_gauss_taps' always are: (j-R)^2 and the normalising sum are the same for both).  out or dog may
 * be NULL.  [zo0, zo1): the planes to produce (zo1 < 0: all of them); planes outside the window are not written, the input
 * is read as far as the filter reaches.  sub (optional): where the 2 x 2 x 2 mean of the produced level goes, a dense
 * (X / 2) x (Y / 2) x (Z / 2) volume; *sub_done says whether this launch wrote it (see HAS_SUB above). */
hipError_t sift3d_launch_blur_fused(hipStream_t s, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z,
                                    const float *taps, int ntaps, const sift3d_blur_tuning *tune, int64_t zo0, int64_t zo1, float *sub,
                                    int *sub_done)
{
    const int R = ntaps / 2;
    if (sub_done) *sub_done = 0;
    if (zo1 < 0) zo1 = Z; /* the default window: the whole volume */
    if (R < 1 || R > SIFT3D_FAST_MAX_R || ntaps != 2 * R + 1 || X % 4 != 0 || X * Y >= (1ll << 29) || (!out && !dog)) return hipErrorNotSupported;
    if (zo0 < 0 || zo1 > Z || zo1 <= zo0) return hipErrorInvalidValue;
    for (int j = 0; j < R; j++)
        if (__builtin_bit_cast(unsigned, taps[j]) != __builtin_bit_cast(unsigned, taps[2 * R - j])) return hipErrorNotSupported;
    fb_taps2 t;
    for (int i = 0; i < 2 * SIFT3D_FAST_MAX_R + 1; i++) t.f[i] = v2f(i < ntaps ? taps[i] : 0.0f);
    bool ok = false;
    switch (R) {
    case 1: ok = launch_ring<1>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune, sub, sub_done); break;
    case 2: ok = launch_ring<2>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune, sub, sub_done); break;
    case 3: ok = launch_ring<3>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune, sub, sub_done); break;
    case 4: ok = launch_ring<4>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune, sub, sub_done); break;
    case 5: ok = launch_ring<5>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune, sub, sub_done); break;
    case 6: ok = launch_ring<6>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune, sub, sub_done); break;
    case 7: ok = launch_ring<7>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune, sub, sub_done); break;
    default: ok = launch_ring<8>(s, in, out, dog, X, Y, Z, zo0, zo1, t, tune, sub, sub_done); break;
    }
    if (!ok) return hipErrorNotSupported;
    return hipGetLastError();
}
