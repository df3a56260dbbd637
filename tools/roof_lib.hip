// tools/roof_lib.hip -- MEASUREMENT AID (not part of the product): the zero-arithmetic march of the fused blur's tiles as a
// library call, so that bench.py can put the ceiling of the kernel's ACCESS SHAPE beside roofline.frac in the line it prints,
// measured in the same process on the same box (round-4 review item 4).  The kernels are tools/stream_roof.hip's k_tile_w:
// a workgroup of 512 threads owns a TW x TH (x, y) tile of an N^3 float volume and marches along z in two chunks, two planes
// of loads in flight per thread, one plane read and one (level only: 8 B/voxel) or two (level + DoG: 12 B/voxel) planes
// stored per step with non-temporal 8-byte stores -- the fused blur's memory traffic with its arithmetic, its LDS and its
// halo removed.  256 workgroups at 512^3: one per CU, as the blur runs.
//
//   int roof_march(int N, int reps, float *ms_out /* 4 */)
//       ms_out[0] 64 x 32 tiles, 1 store stream   ms_out[1] 128 x 16 tiles, 1 store stream
//       ms_out[2] 64 x 32 tiles, 2 store streams  ms_out[3] 128 x 16 tiles, 2 store streams
//   each the BEST of reps launches and of three placements of the buffers: a ceiling is the best a march does, and what it does
//   depends on where its three streams fall in the memory system -- three 2^29-byte buffers back to back put the read and both write
//   streams of a workgroup on the same channels at the same moment (round 5, second half: on one box of the pool the two-store march
//   of back-to-back buffers took 0.35 ms where the blur itself, arithmetic included, took 0.29).
//   returns 0, or a negative number if a HIP call failed (the buffers need 12 N^3 bytes + 48 MB).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int LX, int NW>
__global__ __launch_bounds__(512) void k_march(const float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int X, int Y, int Z, int zlen,
                                               int tiles_x, int tiles_y, long long total)
{
    constexpr int PFD = 2, TW = 2 * LX;
    const int TH = 2 * ((int)blockDim.x / LX);
    const long long lin = blockIdx.x, per = (total + 7) / 8, w = (lin % 8) * per + lin / 8; // XCD x walks the x-th eighth of the tiles
    if (w >= total) return;
    const int tx = (int)(w % tiles_x), ty = (int)((w / tiles_x) % tiles_y), ch = (int)(w / ((long long)tiles_x * tiles_y));
    const int bcp = threadIdx.x % LX, brs = threadIdx.x / LX;
    const long long XY = (long long)X * Y;
    const long long off0 = (long long)(ty * TH + 2 * brs) * X + tx * TW + 2 * bcp, off1 = off0 + X;
    const int z0 = ch * zlen, z1 = z0 + zlen < Z ? z0 + zlen : Z;
    v2f r0[PFD], r1[PFD];
#pragma unroll
    for (int q = 0; q < PFD; q++) {
        const int z = z0 + q < z1 ? z0 + q : z1 - 1;
        r0[q] = *reinterpret_cast<const v2f *>(a + z * XY + off0);
        r1[q] = *reinterpret_cast<const v2f *>(a + z * XY + off1);
    }
    for (int z = z0; z < z1; z += PFD) {
#pragma unroll
        for (int q = 0; q < PFD; q++) {
            if (z + q >= z1) break;
            const v2f p0 = r0[q], p1 = r1[q];
            const int zn = z + q + PFD < z1 ? z + q + PFD : z1 - 1;
            r0[q] = *reinterpret_cast<const v2f *>(a + zn * XY + off0);
            r1[q] = *reinterpret_cast<const v2f *>(a + zn * XY + off1);
            __builtin_nontemporal_store(p0, reinterpret_cast<v2f *>(b + (z + q) * XY + off0));
            __builtin_nontemporal_store(p1, reinterpret_cast<v2f *>(b + (z + q) * XY + off1));
            if (NW > 1) {
                __builtin_nontemporal_store(p0 + p0, reinterpret_cast<v2f *>(c + (z + q) * XY + off0));
                __builtin_nontemporal_store(p1 + p1, reinterpret_cast<v2f *>(c + (z + q) * XY + off1));
            }
        }
    }
}

template <int LX, int NW>
static int run(const float *a, float *b, float *c, int N, int reps, hipEvent_t e0, hipEvent_t e1, float *ms_out)
{
    const int tw = 2 * LX, th = 2 * (512 / LX), nch = 2, zlen = (N + nch - 1) / nch;
    if (N % tw || N % th) return -10;
    const int tiles_x = N / tw, tiles_y = N / th;
    const long long total = (long long)tiles_x * tiles_y * nch, per = (total + 7) / 8;
    std::vector<float> ms;
    for (int r = 0; r < reps + 2; r++) {
        if (hipEventRecord(e0, 0) != hipSuccess) return -3;
        hipLaunchKernelGGL((k_march<LX, NW>), dim3((unsigned)(8 * per)), dim3(512), 0, 0, a, b, c, N, N, N, zlen, tiles_x, tiles_y, total);
        if (hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -4;
        float t = 0;
        if (hipEventElapsedTime(&t, e0, e1) != hipSuccess) return -5;
        if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    if (ms[0] < *ms_out) *ms_out = ms[0];
    return 0;
}

/* detail: 12 floats, [placement][shape] (NULL: not wanted) */
static int roof_march_impl(int N, int reps, float *ms_out, float *detail)
{
    if (N < 128 || reps < 1 || !ms_out) return -1;
    const size_t n = (size_t)N * N * N;
    /* placements of the two output buffers behind the input: back to back, and shifted by amounts that are no multiple of any
     * power-of-two interleave up to 16 MB (floats) */
    const size_t pads[3] = {0, (1u << 18) + (9u << 10) + 192, (3u << 20) + (37u << 10) + 448};
    const size_t extra = 2 * pads[2] + 1024;
    float *base = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    if (hipMalloc(&base, (3 * n + extra) * 4) != hipSuccess) rc = -2;
    if (!rc && hipMemset(base, 1, (3 * n + extra) * 4) != hipSuccess) rc = -2;
    if (!rc && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = -3;
    for (int i = 0; i < 4; i++) ms_out[i] = 1e30f;
    for (int p = 0; p < 3 && !rc; p++) {
        float *a = base, *b = base + n + pads[p], *c = base + 2 * n + 2 * pads[p];
        float one[4] = {1e30f, 1e30f, 1e30f, 1e30f};
        if (!rc) rc = run<32, 1>(a, b, c, N, reps, e0, e1, one + 0);
        if (!rc) rc = run<64, 1>(a, b, c, N, reps, e0, e1, one + 1);
        if (!rc) rc = run<32, 2>(a, b, c, N, reps, e0, e1, one + 2);
        if (!rc) rc = run<64, 2>(a, b, c, N, reps, e0, e1, one + 3);
        for (int i = 0; i < 4; i++) {
            if (one[i] < ms_out[i]) ms_out[i] = one[i];
            if (detail) detail[p * 4 + i] = one[i];
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(base);
    return rc;
}

extern "C" int roof_march(int N, int reps, float *ms_out) { return roof_march_impl(N, reps, ms_out, nullptr); }
extern "C" int roof_march_detail(int N, int reps, float *ms_out, float *detail12) { return roof_march_impl(N, reps, ms_out, detail12); }
