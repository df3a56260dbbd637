/*
 * kernels_volume.hip -- whole-volume kernels of the scale-space pyramid for
 * gfx950 (MI355X): separable Gaussian passes, fused DoG store, 2x2x2
 * subsample, size doubling/halving, 80-neighbour extrema detection.
 *
 * Arithmetic contract (bit-exact with the reference's CPU path,
 * R/src_common/GaussBlur3D.cpp:43-61,329-479 where R/ =
 * /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/): every output
 * is  acc = 0; for j ascending: acc = acc + f[j]*in[c+j-R]  with the multiply
 * and the add rounded separately (this file is compiled -ffp-contract=off),
 * zero outside the volume, float32 intermediates between the x, y and z
 * passes.  A tap that falls outside contributes f*0 = +0, which never changes
 * a float accumulator that started at +0, so feeding zeros and skipping are
 * the same thing.
 *
 * All kernels are HBM-bound streaming kernels: lanes run along x (the fastest
 * axis) with 16-byte accesses whenever the row length allows it; nothing here
 * is GEMM-shaped, so no MFMA.
 */
#include <cstdlib>
#include <type_traits>

#include "sift3d_internal.h"

typedef float v4f __attribute__((ext_vector_type(4)));

template <int VEC>
struct vecT;
template <>
struct vecT<4> {
    typedef v4f type;
};
template <>
struct vecT<1> {
    typedef float type;
};

template <int VEC>
__device__ __forceinline__ typename vecT<VEC>::type vload(const float *p)
{
    return *reinterpret_cast<const typename vecT<VEC>::type *>(p);
}
template <int VEC>
__device__ __forceinline__ void vstore(float *p, typename vecT<VEC>::type v)
{
    *reinterpret_cast<typename vecT<VEC>::type *>(p) = v;
}

/* ------------------------------------------------------------------------ */
/* x pass: one wavefront per 64*VEC-float row segment; the segment and its   */
/* halo are staged in LDS with coalesced 16-byte reads, each lane then pulls */
/* its 4+2R-float window with aligned ds_read_b128 and keeps it in registers.*/
/* ------------------------------------------------------------------------ */
template <int R, int VEC>
__global__ __launch_bounds__(256) void blur_x_kernel(const float *__restrict__ in, float *__restrict__ out, int X,
                                                     long long rows, int segs_per_row, sift3d_taps t)
{
    constexpr int HALO = (VEC == 4) ? ((R + 3) / 4) * 4 : R;
    constexpr int SEG = 64 * VEC;
    constexpr int LROW = SEG + 2 * HALO;
    __shared__ __attribute__((aligned(16))) float lds[4 * LROW];
    typedef typename vecT<VEC>::type V;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long w = (long long)blockIdx.x * 4 + wave;
    const long long nw = rows * segs_per_row;
    float *my = lds + wave * LROW;
    const bool live = w < nw;
    const long long row = live ? w / segs_per_row : 0;
    const int seg = live ? (int)(w % segs_per_row) : 0;
    const int x0 = seg * SEG;
    const float *src = in + row * X;
    const int xs = x0 + lane * VEC;

    V zero = V(0.0f);
    V m = (live && xs < X) ? vload<VEC>(src + xs) : zero;
    vstore<VEC>(my + HALO + lane * VEC, m);
    constexpr int HV = HALO / VEC; /* halo vectors per side */
    if (lane < HV) {
        int xl = x0 - HALO + lane * VEC;
        V h = (live && xl >= 0) ? vload<VEC>(src + xl) : zero;
        vstore<VEC>(my + lane * VEC, h);
    } else if (lane < 2 * HV) {
        int k = lane - HV;
        int xr = x0 + SEG + k * VEC;
        V h = (live && xr < X) ? vload<VEC>(src + xr) : zero;
        vstore<VEC>(my + HALO + SEG + k * VEC, h);
    }
    __syncthreads();

    float win[VEC + 2 * HALO];
#pragma unroll
    for (int i = 0; i < (VEC + 2 * HALO) / VEC; i++) {
        V v = vload<VEC>(my + lane * VEC + i * VEC);
        if constexpr (VEC == 4) {
            win[4 * i + 0] = v.x; win[4 * i + 1] = v.y; win[4 * i + 2] = v.z; win[4 * i + 3] = v.w;
        } else {
            win[i] = v;
        }
    }
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 2 * R + 1; j++) acc = acc + t.f[j] * win[HALO - R + e + j];
        o[e] = acc;
    }
    if (live && xs < X) {
        if constexpr (VEC == 4) {
            v4f r;
            r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
            vstore<4>(out + row * X + xs, r);
        } else {
            out[row * X + xs] = o[0];
        }
    }
}

/* ------------------------------------------------------------------------ */
/* y / z pass: lanes along x, each lane marches along the filtered axis and  */
/* keeps the 2R+1 partial sums that are alive in registers.  Input row q of  */
/* the chunk feeds tap j = q - o of output o, so every output receives its   */
/* taps in ascending order as the rows stream past; each input row is read   */
/* once per chunk.  The loop is unrolled by U = 2R+1 so that the slot of     */
/* every partial sum is a compile-time register.  With DOG the finished row  */
/* is stored together with prev - row (the fused subtract-and-store).        */
/* ------------------------------------------------------------------------ */
/* one group of U = 2R+1 consecutive input rows q0 .. q0+U-1 of the chunk.
 * PIPE (full chunks only): software pipeline one group deep.  nxt[s] holds the input row of step s,
 * loaded while the previous group was being computed, and is refilled with the row of step s of the
 * NEXT group right after use; pvn[s] does the same for the `prev` row of the fused DoG.  Every
 * wavefront therefore keeps U (2U with DoG) 16-byte loads per lane in flight, which is what hides the
 * HBM latency at 2 waves per SIMD. */
template <int R, int VEC, bool DOG, bool FIRST, bool FULL>
__device__ __forceinline__ void col_group(typename vecT<VEC>::type (&acc)[2 * R + 1],
                                          typename vecT<VEC>::type (&nxt)[2 * R + 1],
                                          typename vecT<VEC>::type (&pvn)[2 * R + 1], const float *__restrict__ src,
                                          float *__restrict__ out, const float *__restrict__ prev,
                                          float *__restrict__ dog, long long base, long long S, int L, int CH, int c0,
                                          int q0, const sift3d_taps &t)
{
    constexpr int U = 2 * R + 1;
    typedef typename vecT<VEC>::type V;
#pragma unroll
    for (int s = 0; s < U; s++) {
        const int q = q0 + s;
        const int p = c0 - R + q;
        V v;
        if (FULL) {
            v = nxt[s];
            const int pn = p + U; /* same step of the next group */
            const int pnc = pn < 0 ? 0 : (pn >= L ? L - 1 : pn);
            nxt[s] = vload<VEC>(src + (long long)pnc * S);
        } else {
            const int pc = p < 0 ? 0 : (p >= L ? L - 1 : p);
            v = vload<VEC>(src + (long long)pc * S);
        }
        if (!(p >= 0 && p < L)) v = V(0.0f); /* zero border */
#pragma unroll
        for (int i = 0; i < U; i++) {
            const int j = (s - i + U) % U;
            if (j == 0)
                acc[i] = V(0.0f) + V(t.f[0]) * v;
            else
                acc[i] = acc[i] + V(t.f[j]) * v;
        }
        const int ic = (s + 1) % U; /* slot whose tap 2R was just added */
        const int o = q - 2 * R;
        const int y = c0 + o;
        bool st;
        if (FULL) st = !FIRST || s == 2 * R; /* compile-time: every output of a full chunk is inside the volume */
        else st = (o >= 0 && o < CH && y < L);
        if (st) {
            const long long off = base + (long long)y * S;
            V g = acc[ic];
            vstore<VEC>(out + off, g);
            if constexpr (DOG) {
                V pv = FULL ? pvn[s] : vload<VEC>(prev + off);
                vstore<VEC>(dog + off, pv - g);
            }
        }
        if constexpr (DOG && FULL) {
            int yn = y + U;
            yn = yn < 0 ? 0 : (yn >= L ? L - 1 : yn);
            pvn[s] = vload<VEC>(prev + base + (long long)yn * S);
        }
    }
}

/* FULL: L >= CH and CH + 2R is a multiple of U; the last chunk is shifted back to end at L (its
 * overlap with the previous chunk is recomputed to identical values), so no store needs a test.
 * !FULL: short axes (L < CH): one chunk with tested stores. */
template <int R, int VEC, bool DOG, bool FULL>
__global__ __launch_bounds__(64) void blur_col_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                      const float *__restrict__ prev, float *__restrict__ dog,
                                                      long long nlines, int XV, long long outer_stride, long long S,
                                                      int L, int CH, sift3d_taps t)
{
    constexpr int U = 2 * R + 1;
    typedef typename vecT<VEC>::type V;
    const long long tid = (long long)blockIdx.x * 64 + threadIdx.x;
    if (tid >= nlines) return;
    const long long base = (tid / XV) * outer_stride + (tid % XV) * (long long)VEC;
    int c0 = blockIdx.y * CH;
    if (FULL && c0 > L - CH) c0 = L - CH;
    const int total = CH + 2 * R;
    const float *src = in + base;
    V acc[U], nxt[U], pvn[U];
#pragma unroll
    for (int i = 0; i < U; i++) acc[i] = V(0.0f);
    if (FULL) {
#pragma unroll
        for (int s = 0; s < U; s++) { /* prologue: rows of group 0 */
            const int p = c0 - R + s;
            const int pc = p < 0 ? 0 : (p >= L ? L - 1 : p);
            nxt[s] = vload<VEC>(src + (long long)pc * S);
        }
        if constexpr (DOG) pvn[2 * R] = vload<VEC>(prev + base + (long long)c0 * S);
    }
    col_group<R, VEC, DOG, true, FULL>(acc, nxt, pvn, src, out, prev, dog, base, S, L, CH, c0, 0, t);
    for (int q0 = U; q0 < total; q0 += U)
        col_group<R, VEC, DOG, false, FULL>(acc, nxt, pvn, src, out, prev, dog, base, S, L, CH, c0, q0, t);
}

/* Generic fallback for tap counts outside 3..17 (never used by the pyramid):
 * one thread per voxel, taps from global memory. */
__global__ void blur_axis_generic_kernel(const float *__restrict__ in, float *__restrict__ out, long long X, long long Y,
                                         long long Z, int axis, const float *__restrict__ taps, int ntaps,
                                         const float *__restrict__ prev, float *__restrict__ dog)
{
    const long long n = X * Y * Z;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long x = i % X, y = (i / X) % Y, z = i / (X * Y);
    const long long c = axis == 0 ? x : (axis == 1 ? y : z);
    const long long len = axis == 0 ? X : (axis == 1 ? Y : Z);
    const long long stride = axis == 0 ? 1 : (axis == 1 ? X : X * Y);
    const int h = ntaps / 2;
    float acc = 0.0f;
    for (int j = 0; j < ntaps; j++) {
        long long cc = c + j - h;
        float v = (cc >= 0 && cc < len) ? in[i + (cc - c) * stride] : 0.0f;
        acc = acc + taps[j] * v;
    }
    out[i] = acc;
    if (dog) dog[i] = prev[i] - acc;
}

/* DoG on its own (operator-level API): out = a + (-1)*b, R/src_common/FeatureIO.cpp:1981 */
__global__ void dog_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, long long n4,
                           long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        v4f va = vload<4>(a + 4 * i), vb = vload<4>(b + 4 * i);
        vstore<4>(out + 4 * i, va - vb);
    }
    if (i == 0)
        for (long long k = 4 * n4; k < n; k++) out[k] = a[k] - b[k];
}

/* 2x2x2 mean, association of R/src_common/FeatureIO.cpp:1532-1538:
 * ((p000+p010)+p100)+p110, then + (((p001+p011)+p101)+p111), times 0.125 */
__global__ void subsample_kernel(const float *__restrict__ in, long long X, long long Xl, long long Y, long long Z,
                                 float *__restrict__ out, long long XPout)
{
    /* X: row pitch of in, Xl: its logical row length; XPout: row pitch of out (== Xl / 2 when dense) */
    const long long ox = Xl / 2, oy = Y / 2, oz = Z / 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ox * oy * oz) return;
    const long long x = i % ox, y = (i / ox) % oy, z = i / (ox * oy);
    const float *p0 = in + ((2 * z) * Y + 2 * y) * X + 2 * x;
    const float *p1 = p0 + X * Y;
    float a00, a10, a01, a11, b00, b10, b01, b11;
    if ((X & 1) == 0) {
        float2 r0 = *reinterpret_cast<const float2 *>(p0), r1 = *reinterpret_cast<const float2 *>(p0 + X);
        float2 r2 = *reinterpret_cast<const float2 *>(p1), r3 = *reinterpret_cast<const float2 *>(p1 + X);
        a00 = r0.x; a10 = r0.y; a01 = r1.x; a11 = r1.y;
        b00 = r2.x; b10 = r2.y; b01 = r3.x; b11 = r3.y;
    } else {
        a00 = p0[0]; a10 = p0[1]; a01 = p0[X]; a11 = p0[X + 1];
        b00 = p1[0]; b10 = p1[1]; b01 = p1[X]; b11 = p1[X + 1];
    }
    float s = 0.0f;
    s = s + (((a00 + a01) + a10) + a11);
    s = s + (((b00 + b01) + b10) + b11);
    out[(z * oy + y) * XPout + x] = s * 0.125f;
}

/* Columns [Xl, X) of a pitched volume back to zero (a blur over the pitched width also writes them; "outside the
 * volume" has to read as zero for the next level). */
__global__ void zero_pad_kernel(float *__restrict__ a, float *__restrict__ b, long long X, long long Xl, long long rows)
{
    const long long w = X - Xl;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const long long r = i / w, x = Xl + i % w;
    if (a) a[r * X + x] = 0.0f;
    if (b) b[r * X + x] = 0.0f;
}

/* fioDoubleSize, R/src_common/FeatureIO.cpp:2452-2548: one thread per output
 * voxel (2x,2y,2z)+(dx,dy,dz); edge voxels replicate. */
__global__ void double_size_kernel(const float *__restrict__ in, long long X, long long Y, long long Z,
                                   float *__restrict__ out)
{
    const long long DX = 2 * X, DY = 2 * Y, DZ = 2 * Z;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DX * DY * DZ) return;
    const long long hx = i % DX, hy = (i / DX) % DY, hz = i / (DX * DY);
    const long long x = hx >> 1, y = hy >> 1, z = hz >> 1;
    const int ox = (int)(hx & 1), oy = (int)(hy & 1), oz = (int)(hz & 1);
    float lo[2][2][2];
#pragma unroll
    for (int zz = 0; zz < 2; zz++)
#pragma unroll
        for (int yy = 0; yy < 2; yy++)
#pragma unroll
            for (int xx = 0; xx < 2; xx++) {
                long long sx = x + ((x + xx >= X) ? 0 : xx), sy = y + ((y + yy >= Y) ? 0 : yy),
                          sz = z + ((z + zz >= Z) ? 0 : zz);
                lo[zz][yy][xx] = in[(sz * Y + sy) * X + sx];
            }
    float v;
    const int code = oz * 4 + oy * 2 + ox;
    switch (code) {
    case 0: v = lo[0][0][0]; break;
    case 4: v = 0.5f * (lo[0][0][0] + lo[1][0][0]); break;
    case 2: v = 0.5f * (lo[0][0][0] + lo[0][1][0]); break;
    case 1: v = 0.5f * (lo[0][0][0] + lo[0][0][1]); break;
    case 6: v = 0.25f * (lo[0][0][0] + lo[1][0][0] + lo[0][1][0] + lo[1][1][0]); break;
    case 3: v = 0.25f * (lo[0][0][0] + lo[0][1][0] + lo[0][0][1] + lo[0][1][1]); break;
    case 5: v = 0.25f * (lo[0][0][0] + lo[1][0][0] + lo[0][0][1] + lo[1][0][1]); break;
    default:
        v = 0.125f * (lo[0][0][0] + lo[0][0][1] + lo[0][1][0] + lo[0][1][1] + lo[1][0][0] + lo[1][0][1] + lo[1][1][0] +
                      lo[1][1][1]);
        break;
    }
    out[i] = v;
}

/* fioSubSample2DCenterPixel, R/src_common/FeatureIO.cpp:1670-1714: the eight
 * voxels summed in the order z fastest, then y, then x; divided by 8. */
__global__ void halve_size_kernel(const float *__restrict__ in, long long X, long long Y, long long Z,
                                  float *__restrict__ out)
{
    const long long ox = X / 2, oy = Y / 2, oz = Z / 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ox * oy * oz) return;
    const long long x = i % ox, y = (i / ox) % oy, z = i / (ox * oy);
    const float *p = in + ((2 * z) * Y + 2 * y) * X + 2 * x;
    const long long XY = X * Y;
    float v = 0.0f;
    v = v + p[0];
    v = v + p[XY];
    v = v + p[X];
    v = v + p[XY + X];
    v = v + p[1];
    v = v + p[XY + 1];
    v = v + p[X + 1];
    v = v + p[XY + X + 1];
    out[i] = v / 8.0f;
}

/* ------------------------------------------------------------------------ */
/* Extrema: strict max/min of d_cur over its 26 neighbours, then centre + 26 */
/* of d_prev and (when present) of d_next: the decision of regFindFEATUREIO  */
/* + peak/valleyFunction4D (R/src_common/MultiScale.cpp:2260-2524) followed  */
/* by validateDifferencePeak/Valley3D (:1135-1318).                          */
/*                                                                          */
/* "c > every one of 26 neighbours" == "c > max of the 26", and max/min are  */
/* exact, so the own-level test is done separably: a workgroup owns 64 x by  */
/* EX_ROWS y and marches along z; each wavefront owns one row (two extra     */
/* wavefronts carry the halo rows), gets its x-neighbours with wave-wide DPP */
/* shifts, publishes the row's 3-max / 3-min through LDS, and keeps the 3x3  */
/* plane max/min of planes z-1 and z+1 in registers.  Every voxel of d_cur   */
/* is loaded once; d_prev / d_next are touched only around the rare          */
/* survivors.  Survivors are appended with a wave-aggregated atomic; the     */
/* radix sort restores raster order.                                         */
/* ------------------------------------------------------------------------ */
#define EX_SEGS 64          /* segments of the own-level extrema list (one atomic counter each) */
#define EX_SEG_STRIDE 32    /* counters 32 x 8 bytes apart: different cache lines / L2 channels */
#define EX_ROWS 2          /* output rows per wavefront */
#define EX_STAGE (64 + 8 * 64) /* entries of a wavefront's staging buffer: flushed at 64, one step adds at most 8 per lane */
#define EX_LOAD (EX_ROWS + 2) /* rows loaded per plane (one halo row on each side) */
#define EX_XOUT 248        /* output voxels per wavefront along x: 64 lanes x float4 minus one halo lane each side */
#define EX_RSRC_FLAGS 0x00020000 /* buffer descriptors: raw buffer, 32-bit data format */

__device__ __forceinline__ float dpp_from_lower(float v) /* lane l gets lane l-1 (lane 0 keeps its own) */
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper(float v) /* lane l gets lane l+1 (lane 63 keeps its own) */
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

/* The hardware's own max / min (IEEE mode: a NaN operand yields the other one, as fmaxf / fminf do).  Written as
 * instructions because fmaxf / fminf on values that come straight from memory make the compiler quiet possible signalling
 * NaNs first -- one v_max_f32 v, v, v per loaded element, 12 % of the march loop's vector instructions -- which the
 * instruction does itself. */
__device__ __forceinline__ float ex_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float ex_min(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float ex_max3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float ex_min3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

/* the same shifts with a value of the lane's own for the lane that has no neighbour in the wavefront (lane 0 / lane 63) */
__device__ __forceinline__ float dpp_from_lower_or(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper_or(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

/* max / min over the x-triple of every element of a row: rmax/rmin include the element, l2max/l2min do not */
__device__ __forceinline__ void row_extrema_v(const float (&v)[6], float (&rmax)[4], float (&rmin)[4], float (&l2max)[4], float (&l2min)[4])
{
#pragma unroll
    for (int e = 0; e < 4; e++) {
        l2max[e] = ex_max(v[e], v[e + 2]);
        l2min[e] = ex_min(v[e], v[e + 2]);
        rmax[e] = ex_max3(v[e], v[e + 2], v[e + 1]);
        rmin[e] = ex_min3(v[e], v[e + 2], v[e + 1]);
    }
}
/* a wavefront covers all of its 256 x: the left neighbour of lane 0 and the right neighbour of lane 63 are `edge` */
__device__ __forceinline__ void row_extrema_edge(v4f a, float edge, float (&rmax)[4], float (&rmin)[4], float (&l2max)[4], float (&l2min)[4])
{
    const float v[6] = {dpp_from_lower_or(a.w, edge), a.x, a.y, a.z, a.w, dpp_from_upper_or(a.x, edge)};
    row_extrema_v(v, rmax, rmin, l2max, l2min);
}
__device__ __forceinline__ void row_extrema(v4f a, float (&rmax)[4], float (&rmin)[4], float (&l2max)[4], float (&l2min)[4])
{
    const float v[6] = {dpp_from_lower(a.w), a.x, a.y, a.z, a.w, dpp_from_upper(a.x)};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        l2max[e] = ex_max(v[e], v[e + 2]);
        l2min[e] = ex_min(v[e], v[e + 2]);
        rmax[e] = ex_max3(v[e], v[e + 2], v[e + 1]);
        rmin[e] = ex_min3(v[e], v[e + 2], v[e + 1]);
    }
}

/* The own-level list is cut into EX_SEGS segments, each with its own counter (one returning atomic on a single word
 * saturates near 88 per microsecond chip-wide; the marching kernel appends in batches of 64 or more, so the rate per
 * counter is low).  A segment holds one contiguous range of z: blockIdx.y counts planes / chunks of planes, so the
 * segments taken in order are the volume taken in slabs -- which is what lets the third phase of a lazily evaluated
 * level walk its candidates slab by slab and find the blocks it reads still in the caches. */
__device__ __forceinline__ int ex_segment_of_z_block()
{
    /* fewer z blocks than segments: segment = z block, and the launcher divides the list's capacity by the segments in use */
    return gridDim.y >= EX_SEGS ? (int)(((unsigned long long)blockIdx.y * EX_SEGS) / gridDim.y) : (int)blockIdx.y;
}
static inline int ex_segments_in_use(unsigned z_blocks) { return z_blocks >= EX_SEGS ? EX_SEGS : (int)z_blocks; }

/* First phase.  One wavefront = 248 output voxels along x (64 lanes x float4; the first and last lane
 * only supply x-neighbours) by EX_ROWS rows of ONE plane: it loads EX_ROWS+2 rows of the three planes
 * z-1, z, z+1 as twelve independent 16-byte loads per lane (all in flight together, re-reads are L1/L2
 * hits), reduces them in registers and hands the own-level extrema to the second phase. */
__global__ __launch_bounds__(256) void extrema_kernel(const float *__restrict__ dprev, const float *__restrict__ dcur,
                                                      const float *__restrict__ dnext, int X, int Xl, int Y, int Z, int z_first,
                                                      int z_last, int zchunk, int xtiles,
                                                      sift3d_survivor *__restrict__ surv, unsigned long long *surv_count,
                                                      long long surv_cap)
{
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6); /* wavefront index over (x tile, y tile) */
    const int xt = wv % xtiles, yt = wv / xtiles;
    const int y0 = 1 + yt * EX_ROWS;                 /* first output row */
    if (y0 >= Y - 1) return;
    const int z = z_first + blockIdx.y;
    if (z >= z_last) return;
    const int xv = xt * EX_XOUT - 4 + lane * 4;      /* x of this lane's first element (may be -4 or >= X: clamped loads) */
    const int xld = xv < 0 ? 0 : (xv > X - 4 ? X - 4 : xv);
    const long long XY = (long long)X * Y;
    v4f pl[3][EX_LOAD];
#pragma unroll
    for (int r = 0; r < EX_LOAD; r++) {
        int yy = y0 - 1 + r;
        yy = yy < Y ? yy : Y - 1;
        const long long off = (long long)yy * X + xld;
#pragma unroll
        for (int k = 0; k < 3; k++) pl[k][r] = vload<4>(dcur + (long long)(z - 1 + k) * XY + off);
    }
    float p9max[EX_ROWS][4], p9min[EX_ROWS][4]; /* over the 3x3 of planes z-1 and z+1 together */
    float e8max[EX_ROWS][4], e8min[EX_ROWS][4]; /* over the 8 in-plane neighbours */
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float rmax[EX_LOAD][4], rmin[EX_LOAD][4], l2max[EX_LOAD][4], l2min[EX_LOAD][4];
#pragma unroll
        for (int r = 0; r < EX_LOAD; r++) row_extrema(pl[k][r], rmax[r], rmin[r], l2max[r], l2min[r]);
#pragma unroll
        for (int r = 0; r < EX_ROWS; r++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (k == 1) {
                    e8max[r][e] = fmaxf(fmaxf(rmax[r][e], rmax[r + 2][e]), l2max[r + 1][e]);
                    e8min[r][e] = fminf(fminf(rmin[r][e], rmin[r + 2][e]), l2min[r + 1][e]);
                } else {
                    const float m = fmaxf(fmaxf(rmax[r][e], rmax[r + 1][e]), rmax[r + 2][e]);
                    const float n = fminf(fminf(rmin[r][e], rmin[r + 1][e]), rmin[r + 2][e]);
                    p9max[r][e] = k == 0 ? m : fmaxf(p9max[r][e], m);
                    p9min[r][e] = k == 0 ? n : fminf(p9min[r][e], n);
                }
            }
    }
#pragma unroll
    for (int r = 0; r < EX_ROWS; r++) {
        const int y = y0 + r;
        const v4f cv = pl[1][r + 1];
        const float cc[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float c = cc[e];
            const bool mx = c > fmaxf(e8max[r][e], p9max[r][e]);
            const bool mn = c < fminf(e8min[r][e], p9min[r][e]);
            const int x = xv + e;
            if ((mx || mn) && lane >= 1 && lane <= 62 && x >= 1 && x < Xl - 1 && y < Y - 1) {
                /* own-level extremum: hand it to the second phase.  One returning atomic on a single word
                 * saturates near 88 per microsecond chip-wide, so the list is cut into EX_SEGS segments with
                 * counters 256 bytes apart and a workgroup appends to the segment its index hashes to */
                const int seg = ex_segment_of_z_block();
                const unsigned long long slot = atomicAdd(surv_count + seg * EX_SEG_STRIDE, 1ull);
                if ((long long)slot < surv_cap) {
                    sift3d_survivor sv;
                    sv.idx = (long long)z * XY + (long long)y * X + x;
                    sv.value = c;
                    sv.is_max = mx ? 1 : 0;
                    surv[(long long)seg * surv_cap + (long long)slot] = sv;
                }
            }
        }
    }
}

/* First phase, marching form (used when the volume has enough planes).  A wavefront owns 256 x (64 lanes x float4) by EXM_ROWS rows and walks a chunk of planes; every plane is loaded and
 * reduced once.  Round 3 form.  What a lane carries from plane to plane is, per voxel, four floats:
 *   pm, pn    the 3x3 max / min (centre included) of the plane just below: the "26 neighbours" of the next plane's voxel
 *             that lie in that plane;
 *   cmx, cmn  the voxel's own value if it beat its 8 in-plane neighbours and the plane below (-inf / +inf otherwise): a
 *             candidate waiting for the plane above.
 * A step on plane p computes the 3x3 max m / min n and the 8-neighbour max / min of p, FINISHES plane p-1 (cmx > m: a
 * maximum; cmn < n: a minimum -- a comparison with -inf / +inf is false, so non-candidates need no flag), and restarts the
 * candidates from p.  The decisions are the max / min / compare operations of extrema_kernel on the same values --
 * "c > every one of 26" == "c > max of 8" and "c > max of 9 below" and "c > max of 9 above" -- so the lists are the same.
 * The round-2 form kept the five reduced arrays of three planes (195 registers for two rows); this one keeps four arrays of
 * one plane, which pays for four rows per wavefront (six rows loaded for four instead of four for two: 1.5 instead of 2
 * requests per voxel to L1/L2), two planes of prefetch in registers, and buffer loads (row offset in a VGPR, plane offset in
 * an SGPR: no 64-bit address arithmetic in the loop).  The four wavefronts of a workgroup are neighbours in y, so the halo
 * rows they share are L1 hits. */
#define EXM_ROWS 4 /* by measurement: 2, 3 and 4 rows per wavefront (149 / 196 / 234 registers) take the same time; 4 loads least */
#define EXM_LOAD (EXM_ROWS + 2)
#define EXM_XOUT 256 /* x per wavefront: 64 lanes x float4, every lane an output lane */
#define EXM_STAGE (64 + 4 * 64) /* a wavefront's staging buffer: flushed at 64 after every row, a row adds at most 4 per lane */

__global__ __launch_bounds__(256) void extrema_march_kernel(const float *__restrict__ dcur, int X, int Xl, int Y, int Z, int z_first,
                                                            int z_last, int zchunk, int xtiles, int ygroups,
                                                            sift3d_survivor *__restrict__ surv, unsigned long long *surv_count,
                                                            long long surv_cap)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* wave-uniform by construction: say so, or every buffer descriptor below lands in vector registers */
    /* consecutive workgroups land on consecutive XCDs: XCD x takes the x-th eighth of the (x tile, y group) list, so that the
     * two halo rows a y group shares with each neighbour are fetched into ONE L2 (SIFT3D_MARCH_ORDER 0: the round-3 order) */
#ifndef SIFT3D_MARCH_ORDER
#define SIFT3D_MARCH_ORDER 1
#endif
    const unsigned gx = gridDim.x;
    const unsigned bx = (SIFT3D_MARCH_ORDER && (gx & 7u) == 0) ? (blockIdx.x & 7u) * (gx >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int xt = (int)(bx % (unsigned)xtiles), yg = (int)(bx / (unsigned)xtiles);
    const int y0 = 1 + (yg * 4 + wave) * EXM_ROWS;   /* first output row of this wavefront */
    const int za = z_first + blockIdx.y * zchunk;
    const int zb = za + zchunk < z_last ? za + zchunk : z_last; /* output planes za .. zb-1; plane zb <= Z-1 exists */
    const bool idle = y0 >= Y - 1 || za >= z_last;             /* wave-uniform */
    /* all 64 lanes produce outputs: x = xv .. xv + 3.  The left neighbour of lane 0's first element and the right
     * neighbour of lane 63's last one come from one extra 4-byte load per row in which only those two lanes carry an
     * address inside the buffer (the others, and positions outside the row, are answered with zeros by the bounds check
     * and cost no memory access).  The round-2 form gave up the outer two lanes instead (248 outputs per wavefront): three
     * wavefronts for a 512-voxel row, i.e. 46 % more loads and arithmetic than the row has voxels. */
    const int xv = xt * EXM_XOUT + lane * 4;
    const long long XY = (long long)X * Y;
    unsigned roff[EXM_LOAD], eoff[EXM_LOAD];
    const int xe = lane == 0 ? xv - 1 : (lane == 63 ? xv + 4 : -1);
#pragma unroll
    for (int r = 0; r < EXM_LOAD; r++) {
        int yy = y0 - 1 + r;
        yy = yy < Y ? yy : Y - 1;
        roff[r] = xv < X ? (unsigned)(yy * X + xv) * 4u : 0xFFFFFFFFu; /* X * Y < 2^29: a plane is below 2 GiB; X % 4 == 0 */
        eoff[r] = (xe >= 0 && xe < X) ? (unsigned)(yy * X + xe) * 4u : 0xFFFFFFFFu;
    }
    const int seg = ex_segment_of_z_block();
    /* the chunk's planes za-1 .. zb through one descriptor: (zchunk + 2) planes stay below 4 GiB (the launcher sees to it) */
    const float *const chunk_base = dcur + (long long)(idle ? 0 : za - 1) * XY;
    const int chunk_bytes = idle ? 0 : (int)(unsigned)((long long)(zb - za + 2) * XY * 4);
    const unsigned plane_bytes = (unsigned)(XY * 4);
    /* za-1 <= z.  The plane offset travels in an SGPR, which the hardware's bounds check does not see: a plane past zb
     * (the prefetch runs two planes ahead; its data is never used) goes through a descriptor of no records instead */
    auto load_plane = [&](v4f(&raw)[EXM_LOAD], float(&edge)[EXM_LOAD], int z) {
        const bool ok = z <= zb; /* wave-uniform */
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)chunk_base, 0, ok ? chunk_bytes : 0, EX_RSRC_FLAGS);
        const int so = ok ? (int)((unsigned)(z - (za - 1)) * plane_bytes) : 0;
#pragma unroll
        for (int r = 0; r < EXM_LOAD; r++) raw[r] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)roff[r], so, 0));
#pragma unroll
        for (int r = 0; r < EXM_LOAD; r++) edge[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)eoff[r], so, 0));
    };
    /* Own-level extrema go to a staging buffer of the wavefront in LDS (compacted with a ballot and a prefix count) and
     * from there to the list in batches: ONE returning atomic and a coalesced store per 64 or more of them (a returning
     * atomic per extremum needs s_waitcnt vmcnt(0), which also drains the next plane's loads: 0.33 ms per 512^3 level in
     * round 1).  The order inside the list does not matter: the validated extrema are sorted by key. */
    __shared__ sift3d_survivor stage_all[4][EXM_STAGE];
    sift3d_survivor *const stage = stage_all[wave];
    if (idle) return;
    int pending = 0; /* wave-uniform */
    auto flush = [&]() {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(surv_count + seg * EX_SEG_STRIDE, (unsigned long long)pending);
        base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(base & 0xffffffffull));
        __builtin_amdgcn_wave_barrier(); /* LDS operations of a wavefront execute in issue order: the entries are written */
        for (int i = lane; i < pending; i += 64)
            if ((long long)(base + i) < surv_cap) surv[(long long)seg * surv_cap + (long long)(base + i)] = stage[i];
        __builtin_amdgcn_wave_barrier();
        pending = 0;
    };
    auto stage_hits = [&](bool hit, float c, int is_max, int z, int y, int x) {
        const unsigned long long m = __ballot(hit);
        if (m) { /* wave-uniform */
            if (hit) {
                const int pos = pending + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                sift3d_survivor sv;
                sv.idx = (long long)z * XY + (long long)y * X + x;
                sv.value = c;
                sv.is_max = is_max;
                stage[pos] = sv;
            }
            pending += __popcll(m);
        }
    };
    const float NEG = -__builtin_inff(), POS = __builtin_inff();
    float pm[EXM_ROWS][4], pn[EXM_ROWS][4], cmx[EXM_ROWS][4], cmn[EXM_ROWS][4];
#pragma unroll
    for (int r = 0; r < EXM_ROWS; r++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            pm[r][e] = pn[r][e] = 0.0f;
            cmx[r][e] = NEG; /* plane za-1 has no candidates here: it is the chunk below's, or the volume's face */
            cmn[r][e] = POS;
        }
    /* One plane: reduce it, finish the candidates of the plane below (plane z-1), start this plane's.  FIRST: plane za-1,
     * of which only the 3x3 max / min are wanted. */
    auto step = [&](const v4f(&raw)[EXM_LOAD], const float(&edge)[EXM_LOAD], int z, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        /* a window of three reduced rows (slot = row % 3) and the in-row pair max / min of two (slot = row & 1) */
        float rmax[3][4], rmin[3][4], l2max[2][4], l2min[2][4];
        row_extrema_edge(raw[0], edge[0], rmax[0], rmin[0], l2max[0], l2min[0]);
        row_extrema_edge(raw[1], edge[1], rmax[1], rmin[1], l2max[1], l2min[1]);
#pragma unroll
        for (int r = 0; r < EXM_ROWS; r++) {
            constexpr int dummy = 0;
            (void)dummy;
            const int a = r % 3, b = (r + 1) % 3, c2 = (r + 2) % 3; /* window slots of rows r, r+1 (the output row), r+2 */
            row_extrema_edge(raw[r + 2], edge[r + 2], rmax[c2], rmin[c2], l2max[r & 1], l2min[r & 1]);
            const v4f cv = raw[r + 1];
            const float cc[4] = {cv.x, cv.y, cv.z, cv.w};
            float oldx[4], oldn[4];
            bool hx[4], hn[4], rowhit = false;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float m = ex_max3(rmax[a][e], rmax[b][e], rmax[c2][e]);
                const float n = ex_min3(rmin[a][e], rmin[b][e], rmin[c2][e]);
                if constexpr (!FIRST) {
                    const float e8x = ex_max3(rmax[a][e], rmax[c2][e], l2max[(r + 1) & 1][e]);
                    const float e8n = ex_min3(rmin[a][e], rmin[c2][e], l2min[(r + 1) & 1][e]);
                    oldx[e] = cmx[r][e];
                    oldn[e] = cmn[r][e];
                    hx[e] = oldx[e] > m; /* the candidate of plane z-1 also beats the nine voxels above it */
                    hn[e] = oldn[e] < n;
                    rowhit = rowhit || hx[e] || hn[e];
                    const float c = cc[e];
                    cmx[r][e] = c > ex_max(e8x, pm[r][e]) ? c : NEG;
                    cmn[r][e] = c < ex_min(e8n, pn[r][e]) ? c : POS;
                }
                pm[r][e] = m;
                pn[r][e] = n;
            }
            if constexpr (!FIRST) {
                /* the rare part: extrema of plane z-1 in row y0 + r, inside the searched x and y range */
                const int y = y0 + r;
                const bool rowok = y < Y - 1;
                if (__ballot(rowhit && rowok)) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int x = xv + e;
                        const bool inx = rowok && x >= 1 && x < Xl - 1;
                        /* a voxel is never both: one append serves the maximum and the minimum test */
                        stage_hits((hx[e] || hn[e]) && inx, hx[e] ? oldx[e] : oldn[e], hx[e] ? 1 : 0, z - 1, y, x);
                    }
                    if (pending >= 64) flush();
                }
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    v4f w0[EXM_LOAD], w1[EXM_LOAD], w2[EXM_LOAD];
    float g0[EXM_LOAD], g1[EXM_LOAD], g2[EXM_LOAD];
    load_plane(w0, g0, za - 1);
    load_plane(w1, g1, za);
    load_plane(w2, g2, za + 1);
    step(w0, g0, za - 1, T{});
    /* planes za .. zb (zb only finishes zb-1).  Invariant at the top: w1 holds plane z, w2 plane z+1 (in flight), w0 is free;
     * the three register windows rotate without copies, two planes are always in flight */
    for (int z = za;;) {
        load_plane(w0, g0, z + 2);
        step(w1, g1, z, F{});
        if (++z > zb) break;
        load_plane(w1, g1, z + 2);
        step(w2, g2, z, F{});
        if (++z > zb) break;
        load_plane(w2, g2, z + 2);
        step(w0, g0, z, F{});
        if (++z > zb) break;
    }
    if (pending > 0) flush();
}

/* Second phase: one thread per own-level extremum checks centre + 26 of d_prev and of d_next and
 * appends the survivors as (key, value) pairs.  The list length lives in device memory, so the grid
 * is sized for the capacity and surplus threads leave at once (no host round trip).
 *
 * PAIR: the level below is not stored as a DoG volume; its value at a voxel is gprev_a[i] - gprev_b[i], the two
 * Gaussian levels it is the difference of -- which is how the reference itself validates against a DoG level it never
 * materialises (validateDifferencePeak3D, R/src_common/MultiScale.cpp:1135-1223: fG1 - fG2 at the 27 positions).
 * DEFER: the level above is not stored either, nor is the Gaussian level it would be made from: what passes the test
 * against the level below goes to a second list, and extrema_validate_lazy_kernel evaluates that Gaussian level at the
 * 27 positions around each entry. */
template <bool PAIR, bool DEFER>
__global__ __launch_bounds__(256) void extrema_validate_kernel(const float *__restrict__ dprev, const float *__restrict__ gprev_b,
                                                               const float *__restrict__ dnext,
                                                               int X, int Y, const sift3d_survivor *__restrict__ surv,
                                                               const unsigned long long *__restrict__ surv_count,
                                                               long long surv_cap, unsigned long long *surv_overflow,
                                                               int lvl_id, unsigned long long *__restrict__ keys,
                                                               sift3d_cval *__restrict__ vals, unsigned long long *count,
                                                               long long cap, sift3d_survivor2 *__restrict__ list2,
                                                               unsigned long long *list2_count, long long list2_cap)
{
    const int seg = blockIdx.y; /* surv_cap is the capacity of one segment */
    long long n = (long long)surv_count[seg * EX_SEG_STRIDE];
    if (n > surv_cap) { /* the segment was cut short: tell the host how much room a replay needs */
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(surv_overflow, (unsigned long long)n * gridDim.y); /* gridDim.y = segments in use */
        n = surv_cap;
    }
    /* The grid normally covers the list's capacity (surplus workgroups leave at once); a shorter grid walks the list in
     * strides.  The bound is the same for every thread of the workgroup (the DEFER form votes).  (Round 3 tried a grid sized
     * for 1/256 of the voxels with the strides doing the rest: own-level extrema are 0.4 - 0.5 % of a blob field, so the
     * strides were the common case and every form got slower: 40 -> 65, 24 -> 46, 86 -> 93 us at 512^3.) */
    for (long long i0 = (long long)blockIdx.x * blockDim.x; i0 < n; i0 += (long long)gridDim.x * blockDim.x) {
    const long long i = i0 + threadIdx.x;
    bool ok = i < n;
    if (!DEFER && !ok) continue;
    sift3d_survivor sv;
    sv.idx = 0; sv.value = 0.0f; sv.is_max = 0;
    if (ok) sv = surv[(long long)seg * surv_cap + i];
    const long long XY = (long long)X * Y;
    const float c = sv.value;
    const bool mx = sv.is_max != 0;
    float hval = 0.0f, lval = 0.0f;
    auto prev_at = [&](long long j) -> float { return PAIR ? dprev[j] - gprev_b[j] : dprev[j]; };
    /* Which of the 54 comparisons runs first does not change their conjunction.  The voxel itself in the two neighbour
     * levels is the likeliest to refute an own-level extremum (adjacent DoG levels are strongly correlated there), so it is
     * asked first -- one or two loads per listed voxel, of which there are 0.4 - 0.5 % of the volume -- and the 26 around it
     * only for what is left. */
    if (ok) {
        hval = prev_at(sv.idx);
        ok = mx ? (hval < c) : (hval > c);
        if (!DEFER && dnext) {
            lval = dnext[sv.idx];
            ok = ok && (mx ? (lval < c) : (lval > c));
        }
    }
    if (ok) {
        for (int dz = -1; dz <= 1 && ok; dz++) {
            float q[9];
#pragma unroll
            for (int k = 0; k < 9; k++) q[k] = (dz == 0 && k == 4) ? hval : prev_at(sv.idx + dz * XY + (k / 3 - 1) * X + (k % 3 - 1));
#pragma unroll
            for (int k = 0; k < 9; k++) ok = ok && (mx ? (q[k] < c) : (q[k] > c));
        }
    }
    if constexpr (DEFER) {
        /* one returning atomic per wavefront: the entries of a wavefront go to consecutive slots */
        const unsigned long long m = __ballot(ok);
        if (m == 0) continue;
        const int lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(list2_count + seg, (unsigned long long)__popcll(m));
        const int src = (int)__builtin_ctzll(m);
        base = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(base >> 32), src) << 32) |
               (unsigned)__builtin_amdgcn_readlane((int)(base & 0xffffffffull), src);
        if (!ok) continue;
        const long long slot = (long long)base + __popcll(m & ((1ull << lane) - 1ull));
        if (slot < list2_cap) { /* list2_cap: entries per segment, as for the own-level list this is a subset of: never binds */
            sift3d_survivor2 e; /* the third phase walks its list one entry per wavefront: the divisions are done here, per lane */
            e.x = (int)(sv.idx % X);
            e.y = (int)((sv.idx / X) % Y);
            e.z = (int)(sv.idx / XY);
            e.is_max = mx ? 1 : 0;
            e.value = c;
            e.h = hval;
            list2[(long long)seg * list2_cap + slot] = e;
        }
        continue;
    } else {
        if (ok && dnext) {
            for (int dz = -1; dz <= 1 && ok; dz++) {
                float q[9];
#pragma unroll
                for (int k = 0; k < 9; k++) q[k] = (dz == 0 && k == 4) ? lval : dnext[sv.idx + dz * XY + (k / 3 - 1) * X + (k % 3 - 1)];
#pragma unroll
                for (int k = 0; k < 9; k++) ok = ok && (mx ? (q[k] < c) : (q[k] > c));
            }
        }
        if (!ok) continue;
        const unsigned long long slot = atomicAdd(count, 1ull);
        if ((long long)slot < cap) {
            sift3d_cval r;
            r.value = c;
            r.h = hval;
            r.l = dnext ? lval : 0.0f;
            r.pad = 0.0f;
            keys[slot] = ((unsigned long long)lvl_id << SIFT3D_KEY_LVL_SHIFT) |
                         ((unsigned long long)(mx ? 1 : 0) << SIFT3D_KEY_MAX_SHIFT) | (unsigned long long)sv.idx;
            vals[slot] = r;
        }
    }
    }
}

/* Third phase of a level whose upper neighbour is not stored (the last detection level of an octave): the level above
 * would be D_next = G - blur(G), with blur(G) a Gaussian level nothing else ever reads.  The reference filters the whole
 * volume for it (R/src_common/MultiScale.cpp:405-413) and then looks at 27 voxels around each candidate
 * (validateDifference*3D, :1135-1318); here only those 27 voxels are computed, from the (2R+3)^3 block of G around the
 * candidate, with the operations of the full filter in its order: x pass, y pass, z pass, each output = 0, then
 * + f[j] * input in ascending j with a separate multiply and add, inputs outside the volume read as zero
 * (blur_3d_simpleborders, R/src_common/GaussBlur3D.cpp:329-479) -- the same bits as blur_fused_ring_kernel /
 * blur_x_kernel + blur_col_kernel produce for those voxels.
 *
 * One 64-lane workgroup per candidate, taken from the list in a grid-stride loop (the list length is only known on the
 * device).  The kernel is bound by instruction issue (a 512^3 volume has ~22 000 such candidates on its finest octave,
 * ~150 million filter taps), so: the whole block is requested at once through buffer loads (plane base in the descriptor,
 * 32-bit offsets, out-of-volume lanes and planes answered with zeros by the bounds check -- no address arithmetic, no
 * selects) and consumed as it arrives; the x pass takes TWO planes per step as packed pairs (3 (2R+3) lanes, one 8-byte
 * LDS read + one packed multiply + one packed add per tap); then the y pass over all planes (9 (2R+3) outputs), the z
 * pass (27 outputs) and the comparison on 27 lanes. */
typedef float ex_v2f __attribute__((ext_vector_type(2)));
template <int R>
__global__ __launch_bounds__(64) void extrema_validate_lazy_kernel(const float *__restrict__ g, int X, int Xl, int Y, int Z,
                                                                  const sift3d_survivor2 *__restrict__ list,
                                                                  const unsigned long long *__restrict__ list_count,
                                                                  long long list_cap, int lvl_id,
                                                                  unsigned long long *__restrict__ keys,
                                                                  sift3d_cval *__restrict__ vals, unsigned long long *count,
                                                                  long long cap, sift3d_taps t)
{
    constexpr int U = 2 * R + 1, W = 2 * R + 3, PL = W * W, NLD = (PL + 63) / 64, NP = (W + 1) / 2;
    static_assert(3 * W <= 64, "the x pass of a plane pair fits one wavefront");
    __shared__ ex_v2f raw[2][PL];
    __shared__ float t1[2 * NP * W * 3]; /* [plane][row][dx] (one spare plane: W is odd) */
    __shared__ float t2[W * 9];          /* [plane][dy][dx] */
    const int lane = threadIdx.x;
    /* the list comes in EX_SEGS segments of list_cap entries, one per slab of z (lane = segment): position p of the
     * whole list, segments in order, is entry p - before[s] of the segment s with before[s] <= p < before[s] + len[s] */
    long long len = (long long)list_count[lane];
    len = len < list_cap ? len : list_cap;
    long long before = len; /* inclusive prefix sum over the 64 lanes */
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long up = __shfl_up(before, d, 64);
        if (lane >= d) before += up;
    }
    const long long n = __shfl(before, 63, 64);
    before -= len;
    const long long XY = (long long)X * Y;
    const int plane_bytes = (int)(XY * 4); /* the launcher keeps X * Y below 2^29 */
    /* Which candidates a workgroup takes.  The list is in slabs of z (its segments, in order), and consecutive workgroups
     * land on consecutive XCDs: handing out candidate i to workgroup i mod gridDim spreads every slab over all eight L2s,
     * each of which then fetches the same lines of the level.  Instead XCD x (workgroups x, x + 8, ...) walks the x-th
     * eighth of the list in order: its L2 holds one slab's neighbourhood at a time.  (SIFT3D_LAZY_ORDER 0: the round-2 order.) */
#ifndef SIFT3D_LAZY_ORDER
#define SIFT3D_LAZY_ORDER 1
#endif
    const long long nxcd = (gridDim.x & 7u) == 0 && SIFT3D_LAZY_ORDER ? 8 : 1;
    const long long share = (n + nxcd - 1) / nxcd, first_i = (long long)(blockIdx.x % nxcd) * share;
    const long long last_i = first_i + share < n ? first_i + share : n;
    for (long long i = first_i + blockIdx.x / nxcd; i < last_i; i += gridDim.x / nxcd) {
        const int sg = __popcll(__ballot(before <= i)) - 1; /* before[] ascends: the last segment that starts at or before i */
        const long long first = __shfl(before, sg, 64);
        const sift3d_survivor2 e = list[(long long)sg * list_cap + (i - first)];
        const int x = __builtin_amdgcn_readfirstlane(e.x), y = __builtin_amdgcn_readfirstlane(e.y), z = __builtin_amdgcn_readfirstlane(e.z);
        const bool mx = __builtin_amdgcn_readfirstlane(e.is_max) != 0;
        const float c = e.value;
        /* byte offsets of this lane's elements inside a plane; outside the volume: beyond any record count */
        unsigned eoff[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int el = lane + 64 * k;
            const int gy = y + el / W - (R + 1), gx = x + el % W - (R + 1);
            eoff[k] = (el < PL && gy >= 0 && gy < Y && gx >= 0 && gx < Xl) ? (unsigned)(gy * X + gx) * 4u : 0xFFFFFFFFu;
        }
        float nx[2 * NP][NLD];
#pragma unroll
        for (int pz = 0; pz < 2 * NP; pz++) {
            const int gz = z + pz - (R + 1);
            const bool zin = pz < W && gz >= 0 && gz < Z; /* wave-uniform */
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void *)(g + (zin ? (long long)gz * XY : 0ll)), 0, zin ? plane_bytes : 0, EX_RSRC_FLAGS);
#pragma unroll
            for (int k = 0; k < NLD; k++) nx[pz][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)eoff[k], 0, 0));
        }
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
            ex_v2f *rb = raw[pp & 1]; /* two buffers: the x pass of a pair overlaps the arrival of the next */
#pragma unroll
            for (int k = 0; k < NLD; k++)
                if (lane + 64 * k < PL) {
                    ex_v2f v;
                    v.x = nx[2 * pp][k];
                    v.y = nx[2 * pp + 1][k];
                    rb[lane + 64 * k] = v;
                }
            __syncthreads();
            if (lane < 3 * W) {
                const int row = lane / 3, dx = lane % 3;
                ex_v2f acc = ex_v2f(0.0f);
#pragma unroll
                for (int j = 0; j < U; j++) acc = acc + ex_v2f(t.f[j]) * rb[row * W + dx + j];
                t1[((2 * pp) * W + row) * 3 + dx] = acc.x;
                t1[((2 * pp + 1) * W + row) * 3 + dx] = acc.y;
            }
        }
        __syncthreads();
        for (int o = lane; o < W * 9; o += 64) {
            const int pz = o / 9, dy = (o / 3) % 3, dx = o % 3;
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < U; j++) acc = acc + t.f[j] * t1[(pz * W + dy + j) * 3 + dx];
            t2[o] = acc;
        }
        __syncthreads();
        bool ok = true;
        float dcen = 0.0f;
        const long long idx = (long long)z * XY + (long long)y * X + x;
        if (lane < 27) {
            const int dz = lane / 9, dy = (lane / 3) % 3, dx = lane % 3;
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < U; j++) acc = acc + t.f[j] * t2[(dz + j) * 9 + dy * 3 + dx];
            const float d = g[idx + (long long)(dz - 1) * XY + (long long)(dy - 1) * X + (dx - 1)] - acc;
            ok = mx ? (d < c) : (d > c);
            dcen = d;
        }
        const bool all = __ballot(!ok) == 0ull;
        const float lval = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dcen), 13));
        if (all && lane == 0) {
            const unsigned long long slot = atomicAdd(count, 1ull);
            if ((long long)slot < cap) {
                sift3d_cval r;
                r.value = c;
                r.h = e.h;
                r.l = lval;
                r.pad = 0.0f;
                keys[slot] = ((unsigned long long)lvl_id << SIFT3D_KEY_LVL_SHIFT) |
                             ((unsigned long long)(mx ? 1 : 0) << SIFT3D_KEY_MAX_SHIFT) | (unsigned long long)idx;
                vals[slot] = r;
            }
        }
        __syncthreads(); /* raw / t1 / t2 are free for the next candidate */
    }
}

/* Fallback for row lengths that are not a multiple of 4 (no aligned 16-byte rows) and for tiny volumes:
 * lanes along x, 27 direct loads per voxel, wavefront-wide early-out.  The body serves one detection level; the second
 * kernel below runs the three detection levels of an octave in one launch (blockIdx.z = level * planes + plane). */
__device__ __forceinline__ void extrema_generic_body(const float *__restrict__ dprev, const float *__restrict__ dcur,
                                                     const float *__restrict__ dnext, int X, int Xl, int Y, int z, int lvl_id,
                                                     unsigned long long *__restrict__ keys, sift3d_cval *__restrict__ vals,
                                                     unsigned long long *count, long long cap);

__global__ __launch_bounds__(256) void extrema_generic_kernel(const float *__restrict__ dprev, const float *__restrict__ dcur,
                                                      const float *__restrict__ dnext, int X, int Xl, int Y, int Z, int z_first,
                                                      int lvl_id, unsigned long long *__restrict__ keys,
                                                      sift3d_cval *__restrict__ vals, unsigned long long *count,
                                                      long long cap)
{
    extrema_generic_body(dprev, dcur, dnext, X, Xl, Y, (int)blockIdx.z + z_first, lvl_id, keys, vals, count, cap);
}

struct ex_octave5 {
    const float *d[5]; /* the five DoG levels of an octave: detection level l tests d[l + 1] against d[l] and d[l + 2] */
};
__global__ __launch_bounds__(256) void extrema_generic_octave_kernel(ex_octave5 o, int X, int Xl, int Y, int Z, int lvl_id0,
                                                                     unsigned long long *__restrict__ keys,
                                                                     sift3d_cval *__restrict__ vals, unsigned long long *count,
                                                                     long long cap)
{
    const int planes = Z - 2, l = (int)blockIdx.z / planes, z = 1 + (int)blockIdx.z % planes;
    extrema_generic_body(o.d[l], o.d[l + 1], o.d[l + 2], X, Xl, Y, z, lvl_id0 + l, keys, vals, count, cap);
}

__device__ __forceinline__ void extrema_generic_body(const float *__restrict__ dprev, const float *__restrict__ dcur,
                                                     const float *__restrict__ dnext, int X, int Xl, int Y, int z, int lvl_id,
                                                     unsigned long long *__restrict__ keys, sift3d_cval *__restrict__ vals,
                                                     unsigned long long *count, long long cap)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool inside = (x >= 1 && x < Xl - 1 && y >= 1 && y < Y - 1);
    const long long XY = (long long)X * Y;
    const long long idx = (long long)z * XY + (long long)y * X + x;
    bool mx = inside, mn = inside;
    float c = 0.0f;
    if (inside) c = dcur[idx];
#pragma unroll
    for (int dz = -1; dz <= 1; dz++) {
        if (inside) {
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++) {
                    if (dz == 0 && dy == 0 && dx == 0) continue;
                    float v = dcur[idx + dz * XY + dy * X + dx];
                    mx = mx && (v < c);
                    mn = mn && (v > c);
                }
        }
        if (!__any(mx || mn)) return;
    }
    if (mx || mn) {
        const float *lv[2] = {dprev, dnext};
        for (int l = 0; l < 2; l++) {
            const float *d = lv[l];
            if (!d) continue;
            for (int dz = -1; dz <= 1 && (mx || mn); dz++)
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        float v = d[idx + dz * XY + dy * X + dx];
                        mx = mx && (v < c);
                        mn = mn && (v > c);
                    }
        }
    }
    if (mx || mn) {
        unsigned long long slot = atomicAdd(count, 1ull);
        if ((long long)slot < cap) {
            sift3d_cval r;
            r.value = c;
            r.h = dprev[idx];
            r.l = dnext ? dnext[idx] : 0.0f;
            r.pad = 0.0f;
            keys[slot] = ((unsigned long long)lvl_id << SIFT3D_KEY_LVL_SHIFT) |
                         ((unsigned long long)(mx ? 1 : 0) << SIFT3D_KEY_MAX_SHIFT) | (unsigned long long)idx;
            vals[slot] = r;
        }
    }
}


/* ------------------------------------------------------------------------ */
/* launchers                                                                */
/* ------------------------------------------------------------------------ */
static inline sift3d_taps pack_taps(const float *taps, int n)
{
    sift3d_taps t;
    for (int i = 0; i < 2 * SIFT3D_FAST_MAX_R + 1; i++) t.f[i] = i < n ? taps[i] : 0.0f;
    return t;
}

template <int R, int VEC>
static void launch_x(hipStream_t s, const float *in, float *out, int64_t X, int64_t rows, const sift3d_taps &t)
{
    const int seg = 64 * VEC;
    const int spr = (int)((X + seg - 1) / seg);
    const long long nw = rows * spr;
    const unsigned blocks = (unsigned)((nw + 3) / 4);
    hipLaunchKernelGGL((blur_x_kernel<R, VEC>), dim3(blocks), dim3(256), 0, s, in, out, (int)X, (long long)rows, spr, t);
}

template <int R>
static void dispatch_x(hipStream_t s, const float *in, float *out, int64_t X, int64_t rows, const sift3d_taps &t)
{
    if (X % 4 == 0) launch_x<R, 4>(s, in, out, X, rows, t);
    else launch_x<R, 1>(s, in, out, X, rows, t);
}

#define SIFT3D_R_SWITCH(R_, CALL)            \
    switch (R_) {                            \
    case 1: { constexpr int RR = 1; CALL; } break; \
    case 2: { constexpr int RR = 2; CALL; } break; \
    case 3: { constexpr int RR = 3; CALL; } break; \
    case 4: { constexpr int RR = 4; CALL; } break; \
    case 5: { constexpr int RR = 5; CALL; } break; \
    case 6: { constexpr int RR = 6; CALL; } break; \
    case 7: { constexpr int RR = 7; CALL; } break; \
    case 8: { constexpr int RR = 8; CALL; } break; \
    default: break;                          \
    }

static void launch_generic(hipStream_t s, const float *in, float *out, int64_t X, int64_t Y, int64_t Z, int axis,
                           const float *d_taps, int ntaps, const float *prev, float *dog)
{
    const long long n = X * Y * Z;
    hipLaunchKernelGGL(blur_axis_generic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out,
                       (long long)X, (long long)Y, (long long)Z, axis, d_taps, ntaps, prev, dog);
}

hipError_t sift3d_launch_blur_x(hipStream_t s, const float *in, float *out, int64_t X, int64_t Y, int64_t Z,
                                const float *taps, int ntaps, const float *d_taps)
{
    const int R = ntaps / 2;
    if (R >= 1 && R <= SIFT3D_FAST_MAX_R) {
        sift3d_taps t = pack_taps(taps, ntaps);
        SIFT3D_R_SWITCH(R, (dispatch_x<RR>(s, in, out, X, Y * Z, t)));
    } else {
        launch_generic(s, in, out, X, Y, Z, 0, d_taps, ntaps, nullptr, nullptr);
    }
    return hipGetLastError();
}

/* chunk length along the marched axis: k*U - 2R outputs (so that the chunk is a whole number of
 * U-row groups), k chosen so that the grid has a few thousand wavefronts while the 2R lead-in rows
 * stay a small fraction of the chunk */
static inline int chunk_len(int R, int64_t L, long long waves_x)
{
    const int U = 2 * R + 1;
    /* cost of a choice = rows streamed (outputs + 2R lead-in rows per chunk + the overlap of the shifted
     * last chunk), inflated when the grid has fewer than ~2048 wavefronts (8 per CU) to hide latency */
    int best = 0;
    double best_cost = 0;
    for (int k = 2; k * U - 2 * R <= L; k++) {
        const int ch = k * U - 2 * R;
        const long long chunks = (L + ch - 1) / ch;
        const double waves = (double)chunks * (double)waves_x;
        double cost = (double)chunks * (ch + 2 * R);
        if (waves < 2048.0) cost *= 1.0 + 0.6 * (2048.0 - waves) / 2048.0;
        if (best == 0 || cost < best_cost) {
            best = ch;
            best_cost = cost;
        }
    }
    return best; /* 0: the axis is shorter than the smallest full chunk */
}

template <int R, int VEC, bool DOG>
static void launch_col(hipStream_t s, const float *in, float *out, const float *prev, float *dog, long long nlines, int XV,
                       long long outer_stride, long long S, int L, const sift3d_taps &t)
{
    constexpr int U = 2 * R + 1;
    const long long wx = (nlines + 63) / 64;
    int ch = chunk_len(R, L, wx);
    if (ch > 0) {
        const unsigned chunks = (unsigned)((L + ch - 1) / ch);
        dim3 grid((unsigned)wx, chunks);
        hipLaunchKernelGGL((blur_col_kernel<R, VEC, DOG, true>), grid, dim3(64), 0, s, in, out, prev, dog, nlines, XV,
                           outer_stride, S, L, ch, t);
    } else {
        ch = ((L + 2 * R + U - 1) / U) * U - 2 * R; /* one chunk covering the axis */
        dim3 grid((unsigned)wx, 1);
        hipLaunchKernelGGL((blur_col_kernel<R, VEC, DOG, false>), grid, dim3(64), 0, s, in, out, prev, dog, nlines, XV,
                           outer_stride, S, L, ch, t);
    }
}

hipError_t sift3d_launch_blur_y(hipStream_t s, const float *in, float *out, int64_t X, int64_t Y, int64_t Z,
                                const float *taps, int ntaps, const float *d_taps)
{
    const int R = ntaps / 2;
    if (R >= 1 && R <= SIFT3D_FAST_MAX_R) {
        sift3d_taps t = pack_taps(taps, ntaps);
        if (X % 4 == 0) {
            const int XV = (int)(X / 4);
            SIFT3D_R_SWITCH(R, (launch_col<RR, 4, false>(s, in, out, nullptr, nullptr, (long long)XV * Z, XV, X * Y, X, (int)Y, t)));
        } else {
            SIFT3D_R_SWITCH(R, (launch_col<RR, 1, false>(s, in, out, nullptr, nullptr, (long long)X * Z, (int)X, X * Y, X, (int)Y, t)));
        }
    } else {
        launch_generic(s, in, out, X, Y, Z, 1, d_taps, ntaps, nullptr, nullptr);
    }
    return hipGetLastError();
}

hipError_t sift3d_launch_blur_z(hipStream_t s, const float *in, float *out, const float *prev, float *dog, int64_t X,
                                int64_t Y, int64_t Z, const float *taps, int ntaps, const float *d_taps)
{
    const int R = ntaps / 2;
    const long long XY = X * Y;
    if (R >= 1 && R <= SIFT3D_FAST_MAX_R && XY < (1ll << 31)) {
        sift3d_taps t = pack_taps(taps, ntaps);
        if (XY % 4 == 0) {
            const long long nl = XY / 4;
            if (dog) {
                SIFT3D_R_SWITCH(R, (launch_col<RR, 4, true>(s, in, out, prev, dog, nl, (int)nl, 0, XY, (int)Z, t)));
            } else {
                SIFT3D_R_SWITCH(R, (launch_col<RR, 4, false>(s, in, out, nullptr, nullptr, nl, (int)nl, 0, XY, (int)Z, t)));
            }
        } else {
            if (dog) {
                SIFT3D_R_SWITCH(R, (launch_col<RR, 1, true>(s, in, out, prev, dog, XY, (int)XY, 0, XY, (int)Z, t)));
            } else {
                SIFT3D_R_SWITCH(R, (launch_col<RR, 1, false>(s, in, out, nullptr, nullptr, XY, (int)XY, 0, XY, (int)Z, t)));
            }
        }
    } else {
        launch_generic(s, in, out, X, Y, Z, 2, d_taps, ntaps, prev, dog);
    }
    return hipGetLastError();
}

hipError_t sift3d_launch_dog(hipStream_t s, const float *a, const float *b, float *out, int64_t n)
{
    const long long n4 = n / 4;
    const long long th = n4 > 0 ? n4 : 1;
    hipLaunchKernelGGL(dog_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, a, b, out, n4, (long long)n);
    return hipGetLastError();
}

hipError_t sift3d_launch_subsample(hipStream_t s, const float *in, int64_t X, int64_t Xl, int64_t Y, int64_t Z, float *out,
                                   int64_t XPout)
{
    const long long n = (Xl / 2) * (Y / 2) * (Z / 2);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(subsample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, (long long)X, (long long)Xl,
                       (long long)Y, (long long)Z, out, (long long)XPout);
    return hipGetLastError();
}

/* The coarsest octaves (at most 4096 voxels: 16^3 and below) are launch latency and nothing else: fifteen blur launches
 * for a few microseconds of work.  One 1024-thread workgroup keeps the octave in LDS and produces its five levels and
 * five DoGs: per level x, y, z pass (the arithmetic of filter_1d as everywhere: ascending taps, separate multiply and
 * add, taps outside the volume skipped = adding the +0 the zero border contributes), then D = L_prev - L_new. */
template <int AXIS>
__device__ __forceinline__ void tiny_pass(const float *src, float *dst, int X, int Y, int Z, int N, const float *f, int nt)
{
    const int R = nt / 2;
    const int len = AXIS == 0 ? X : (AXIS == 1 ? Y : Z);
    const int st = AXIS == 0 ? 1 : (AXIS == 1 ? X : X * Y);
    for (int s = threadIdx.x; s < N; s += 1024) {
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

__global__ __launch_bounds__(1024) void tiny_octave_kernel(const float *__restrict__ L0, sift3d_octave_out o, int X, int XP, int Y, int Z,
                                                           sift3d_octave_taps t)
{
    __shared__ float buf[3][SIFT3D_TINY_VOX];
    __shared__ float taps[2 * SIFT3D_FAST_MAX_R + 1];
    const int N = X * Y * Z;
    float *cur = buf[0], *a = buf[1], *b = buf[2];
    for (int s = threadIdx.x; s < N; s += 1024) cur[s] = L0[(long long)(s / X) * XP + s % X];
    __syncthreads();
    for (int lvl = 0; lvl < 5; lvl++) {
        const int nt = t.n[lvl];
        if (threadIdx.x < nt) taps[threadIdx.x] = t.f[lvl][threadIdx.x];
        __syncthreads();
        tiny_pass<0>(cur, a, X, Y, Z, N, taps, nt);
        tiny_pass<1>(a, b, X, Y, Z, N, taps, nt);
        tiny_pass<2>(b, a, X, Y, Z, N, taps, nt);
        for (int s = threadIdx.x; s < N; s += 1024) {
            const long long g = (long long)(s / X) * XP + s % X; /* pad columns stay zero */
            const float v = a[s];
            if (o.L[lvl]) o.L[lvl][g] = v;
            o.D[lvl][g] = cur[s] - v;
        }
        __syncthreads();
        float *tmp = cur; cur = a; a = tmp;
    }
}

hipError_t sift3d_launch_tiny_octave(hipStream_t s, const float *L0, const sift3d_octave_out &o, int64_t X, int64_t XP, int64_t Y,
                                     int64_t Z, const sift3d_octave_taps &t)
{
    if (X * Y * Z > SIFT3D_TINY_VOX) return hipErrorNotSupported;
    for (int l = 0; l < 5; l++)
        if (t.n[l] < 1 || t.n[l] > 2 * SIFT3D_FAST_MAX_R + 1 || (t.n[l] & 1) == 0 || !o.D[l]) return hipErrorNotSupported;
    hipLaunchKernelGGL(tiny_octave_kernel, dim3(1), dim3(1024), 0, s, L0, o, (int)X, (int)XP, (int)Y, (int)Z, t);
    return hipGetLastError();
}

hipError_t sift3d_launch_zero_pad(hipStream_t s, float *a, float *b, int64_t X, int64_t Xl, int64_t rows)
{
    const long long n = rows * (X - Xl);
    if (n <= 0 || (!a && !b)) return hipSuccess;
    hipLaunchKernelGGL(zero_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, (long long)X, (long long)Xl,
                       (long long)rows);
    return hipGetLastError();
}

hipError_t sift3d_launch_double_size(hipStream_t s, const float *in, int64_t X, int64_t Y, int64_t Z, float *out)
{
    const long long n = 8 * X * Y * Z;
    hipLaunchKernelGGL(double_size_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, (long long)X,
                       (long long)Y, (long long)Z, out);
    return hipGetLastError();
}

hipError_t sift3d_launch_halve_size(hipStream_t s, const float *in, int64_t X, int64_t Y, int64_t Z, float *out)
{
    const long long n = (X / 2) * (Y / 2) * (Z / 2);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(halve_size_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, (long long)X,
                       (long long)Y, (long long)Z, out);
    return hipGetLastError();
}

template <int R>
static void launch_validate_lazy(hipStream_t s, const sift3d_extrema_lazy &lz, int nseg, int64_t X, int64_t Xl, int64_t Y, int64_t Z, int lvl_id,
                                 unsigned long long *keys, sift3d_cval *vals, unsigned long long *count, int64_t cap)
{
    sift3d_taps t;
    for (int i = 0; i < 2 * SIFT3D_FAST_MAX_R + 1; i++) t.f[i] = i < lz.ntaps ? lz.taps[i] : 0.0f;
    /* a grid-stride loop over a list whose length only the device knows: enough single-wavefront workgroups to fill
     * the chip (256 CUs x 16), never more than the list can hold */
    long long wgs = X * Y * Z / 2048; /* the finest octaves fill the chip; a coarse one does not pay for 4096 idle workgroups */
    wgs = wgs < 64 ? 64 : (wgs > 4096 ? 4096 : wgs);
    if (wgs > lz.list2_cap) wgs = lz.list2_cap;
    if (wgs >= 8) wgs = wgs / 8 * 8; /* whole rounds of the eight XCDs: the kernel gives each an eighth of the list */
    hipLaunchKernelGGL(extrema_validate_lazy_kernel<R>, dim3((unsigned)wgs), dim3(64), 0, s, lz.next_g, (int)X, (int)Xl, (int)Y, (int)Z,
                       lz.list2, lz.list2_count, (long long)(lz.list2_cap / nseg), lvl_id, keys, vals, count, (long long)cap, t);
}

/* The three detection levels of an octave of at most SIFT3D_TINY_VOX voxels, all five DoG levels stored, in one launch
 * (the per-level path is two to three launches per level: fifteen small launches at the very end of the pyramid's chain
 * for the three smallest octaves of a 512^3 volume). */
hipError_t sift3d_launch_extrema_octave_small(hipStream_t s, const float *const d[5], int64_t X, int64_t Xl, int64_t Y, int64_t Z,
                                              int lvl_id0, unsigned long long *keys, sift3d_cval *vals, unsigned long long *count,
                                              int64_t cap)
{
    if (Xl < 3 || Y < 3 || Z < 3) return hipSuccess;
    ex_octave5 o;
    for (int i = 0; i < 5; i++) o.d[i] = d[i];
    dim3 grid((unsigned)((X + 63) / 64), (unsigned)((Y + 3) / 4), (unsigned)(3 * (Z - 2)));
    hipLaunchKernelGGL(extrema_generic_octave_kernel, grid, dim3(256), 0, s, o, (int)X, (int)Xl, (int)Y, (int)Z, lvl_id0, keys, vals, count,
                       (long long)cap);
    return hipGetLastError();
}

hipError_t sift3d_launch_extrema(hipStream_t s, const float *dprev, const float *dcur, const float *dnext, int64_t X,
                                 int64_t Xl, int64_t Y, int64_t Z, int z_lo, int z_hi, int lvl_id, unsigned long long *keys,
                                 sift3d_cval *vals, unsigned long long *count, int64_t cap, sift3d_survivor *surv,
                                 unsigned long long *surv_count, unsigned long long *surv_overflow, int64_t surv_cap,
                                 bool zero_counters, const sift3d_extrema_lazy *lazy)
{
    if (Xl < 3 || Y < 3 || Z < 3) return hipSuccess;
    const bool pair = lazy && lazy->prev_b, defer = lazy && lazy->next_g;
    /* X: row pitch (== Xl for a dense volume), Xl: logical row length; interior planes 1..Z-2, further restricted to [z_lo, z_hi) (Z-slab mode keeps only its own slices) */
    const int z0 = z_lo > 1 ? z_lo : 1;
    const int z1 = z_hi < (int)Z - 1 ? z_hi : (int)Z - 1;
    if (z1 <= z0) return hipSuccess;
    if (X % 4 == 0 && X >= 8 && surv && surv_cap > 0) {
        const int xtiles = (int)((X - 2 + EX_XOUT - 1) / EX_XOUT);
        const int ytiles = (int)((Y - 2 + EX_ROWS - 1) / EX_ROWS);
        const long long waves_xy = (long long)xtiles * ytiles;
        if (zero_counters) { /* the pipeline hands every level its own, already zeroed, counter set instead */
            hipError_t e = hipMemsetAsync(surv_count, 0, sizeof(unsigned long long) * EX_SEGS * EX_SEG_STRIDE, s);
            if (e != hipSuccess) return e;
        }
        /* marching form when chunks of >= 8 planes still give the chip a few thousand wavefronts (a wavefront of the march
         * takes EXM_ROWS rows, the four of a workgroup are neighbours in y); its buffer descriptor covers a chunk and the two
         * planes around it, which must stay below 4 GiB */
        const int xtiles_m = (int)((X + EXM_XOUT - 1) / EXM_XOUT);
        const int ytiles_m = (int)((Y - 2 + EXM_ROWS - 1) / EXM_ROWS), ygroups = (ytiles_m + 3) / 4;
        const long long waves_m = (long long)xtiles_m * ytiles_m;
        const long long plane_bytes = X * Y * 4;
        /* the longest chunk that still gives the chip 2 048 wavefronts (two per SIMD: what the kernel's registers allow) --
         * 512^3: 64 planes, 256^3: 8 --, else the longest that gives 512 (128^3: 8); below that the plane-per-block form */
        int zchunk = 1;
        for (int need = 2048; need >= 512 && zchunk == 1; need /= 4)
            for (int zc = 64; zc >= 8; zc /= 2)
                if (waves_m * ((z1 - z0 + zc - 1) / zc) >= need && (zc + 2) * plane_bytes < (1ll << 32)) {
                    zchunk = zc;
                    break;
                }
        const unsigned nz = (unsigned)((z1 - z0 + zchunk - 1) / zchunk);
        const int nseg = ex_segments_in_use(nz);
        const long long segcap = surv_cap / nseg; /* entries per segment of the own-level list */
        if (zchunk >= 2) {
            dim3 grid((unsigned)(xtiles_m * ygroups), nz);
            hipLaunchKernelGGL(extrema_march_kernel, grid, dim3(256), 0, s, dcur, (int)X, (int)Xl, (int)Y, (int)Z, z0, z1, zchunk, xtiles_m,
                               ygroups, surv, surv_count, segcap);
        } else { /* reads the own level only: its neighbour-level arguments are unused */
            dim3 grid((unsigned)((waves_xy + 3) / 4), nz);
            hipLaunchKernelGGL(extrema_kernel, grid, dim3(256), 0, s, dprev, dcur, dnext, (int)X, (int)Xl, (int)Y, (int)Z, z0, z1, zchunk,
                               xtiles, surv, surv_count, segcap);
        }
        /* the second launch covers the list capacity, reads the true length on the device, and flags an
         * overflow for cand_finalize to widen the list and replay */
        const dim3 vgrid((unsigned)((segcap + 255) / 256), (unsigned)nseg);
#define SIFT3D_VALIDATE(PAIR_, DEFER_)                                                                                                   \
    hipLaunchKernelGGL((extrema_validate_kernel<PAIR_, DEFER_>), vgrid, dim3(256), 0, s, dprev, pair ? lazy->prev_b : nullptr,          \
                       defer ? nullptr : dnext, (int)X, (int)Y, surv, surv_count, segcap, surv_overflow, lvl_id, keys, vals, count,      \
                       (long long)cap, defer ? lazy->list2 : nullptr, defer ? lazy->list2_count : nullptr,                               \
                       (long long)(defer ? lazy->list2_cap / nseg : 0))
        if (pair && defer) SIFT3D_VALIDATE(true, true);
        else if (pair) SIFT3D_VALIDATE(true, false);
        else if (defer) SIFT3D_VALIDATE(false, true);
        else SIFT3D_VALIDATE(false, false);
#undef SIFT3D_VALIDATE
        if (defer) {
            const int R = lazy->ntaps / 2;
            if (lazy->ntaps != 2 * R + 1 || R < 1 || R > SIFT3D_FAST_MAX_R || !lazy->list2 || !lazy->list2_count || lazy->list2_cap <= 0 ||
                X * Y >= (1ll << 29))
                return hipErrorInvalidValue;
            /* the level above the last detection level is always the 17-tap one (sigma 3.09: the schedule of
             * MultiScale.cpp:288-294 does not depend on the input), so that is the one instantiation */
            if (R != SIFT3D_FAST_MAX_R) return hipErrorNotSupported;
            launch_validate_lazy<SIFT3D_FAST_MAX_R>(s, *lazy, nseg, X, Xl, Y, Z, lvl_id, keys, vals, count, cap);
        }
    } else {
        if (pair || defer) return hipErrorNotSupported; /* the caller keeps such shapes on stored DoG levels */
        dim3 grid((unsigned)((X + 63) / 64), (unsigned)((Y + 3) / 4), (unsigned)(z1 - z0));
        hipLaunchKernelGGL(extrema_generic_kernel, grid, dim3(256), 0, s, dprev, dcur, dnext, (int)X, (int)Xl, (int)Y, (int)Z, z0,
                           lvl_id, keys, vals, count, (long long)cap);
    }
    return hipGetLastError();
}
