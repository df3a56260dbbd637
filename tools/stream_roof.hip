// Development aid (GPU box): what the memory system sustains for the traffic mixes of the fused blur, with no arithmetic.
//   copy   : read 4 B, write 4 B per element  (the "float4 copy" the microarchitecture guide quotes at 6.29 TB/s)
//   1r2w   : read 4 B, write 8 B (two output streams: level + DoG)   -- the 12 B/voxel launches
//   1r1w-t : the same 1 read / 2 writes in the blur's own access shape: a workgroup owns a 64 x 32 (x, y) tile of a
//            512^3 volume and marches along z, 256-byte row segments, non-temporal 8-byte stores
// usage: stream_roof [N=512] [reps=20] [stagger_bytes=0]   (stagger: b starts stagger bytes, c 2 x stagger bytes into its allocation)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_copy(const v4f *__restrict__ a, v4f *__restrict__ b, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}
// four independent 16-byte loads per thread in flight before the first store, one-shot grid (what a tuned elementwise kernel does)
__global__ __launch_bounds__(256) void k_copy4(const v4f *__restrict__ a, v4f *__restrict__ b, long long n)
{
    const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
    v4f v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i0 + 256 * u < n) v[u] = a[i0 + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i0 + 256 * u < n) b[i0 + 256 * u] = v[u];
}
__global__ __launch_bounds__(256) void k_1r2w4(const v4f *__restrict__ a, v4f *__restrict__ b, v4f *__restrict__ c, long long n)
{
    const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
    v4f v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i0 + 256 * u < n) v[u] = a[i0 + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i0 + 256 * u < n) { b[i0 + 256 * u] = v[u]; c[i0 + 256 * u] = v[u] + v[u]; }
}
__global__ void k_1r2w(const v4f *__restrict__ a, v4f *__restrict__ b, v4f *__restrict__ c, long long n, int nt)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const v4f v = a[i];
        if (nt) { __builtin_nontemporal_store(v, &b[i]); __builtin_nontemporal_store(v + v, &c[i]); }
        else { b[i] = v; c[i] = v + v; }
    }
}
// tile march: 512 threads, thread = (column pair, row pair) of a 64 x 32 tile, one plane per iteration
__global__ __launch_bounds__(1024) void k_tile(const float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int X, int Y, int Z, int zlen, int tiles_x, int tiles_y, long long total)
{
    const long long lin = blockIdx.x, per = (total + 7) / 8, w = (lin % 8) * per + lin / 8;
    if (w >= total) return;
    const int tx = (int)(w % tiles_x), ty = (int)((w / tiles_x) % tiles_y), ch = (int)(w / ((long long)tiles_x * tiles_y));
    const int bcp = threadIdx.x & 31, brs = threadIdx.x >> 5;
    const long long XY = (long long)X * Y;
    extern __shared__ float dyn[]; if (X < 0) dyn[threadIdx.x] = 0; /* keeps the dynamic LDS request alive */
    const int TYt = 2 * (blockDim.x >> 5);
    const long long off0 = (long long)(ty * TYt + 2 * brs) * X + tx * 64 + 2 * bcp, off1 = off0 + X;
    const int z0 = ch * zlen, z1 = z0 + zlen < Z ? z0 + zlen : Z;
    v2f p0 = *reinterpret_cast<const v2f *>(a + z0 * XY + off0), p1 = *reinterpret_cast<const v2f *>(a + z0 * XY + off1);
    for (int z = z0; z < z1; z++) {
        v2f q0 = p0, q1 = p1;
        if (z + 1 < z1) { q0 = *reinterpret_cast<const v2f *>(a + (z + 1) * XY + off0); q1 = *reinterpret_cast<const v2f *>(a + (z + 1) * XY + off1); }
        __builtin_nontemporal_store(p0, reinterpret_cast<v2f *>(b + z * XY + off0));
        __builtin_nontemporal_store(p1, reinterpret_cast<v2f *>(b + z * XY + off1));
        __builtin_nontemporal_store(p0 + p0, reinterpret_cast<v2f *>(c + z * XY + off0));
        __builtin_nontemporal_store(p1 + p1, reinterpret_cast<v2f *>(c + z * XY + off1));
        p0 = q0; p1 = q1;
    }
}
// the tile march with PFD planes of the tile in flight per thread (a register ring)
template <int PFD>
__global__ __launch_bounds__(512) void k_tile_pf(const float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int X, int Y, int Z, int zlen, int tiles_x, int tiles_y, long long total)
{
    const long long lin = blockIdx.x, per = (total + 7) / 8, w = (lin % 8) * per + lin / 8;
    if (w >= total) return;
    const int tx = (int)(w % tiles_x), ty = (int)((w / tiles_x) % tiles_y), ch = (int)(w / ((long long)tiles_x * tiles_y));
    const int bcp = threadIdx.x & 31, brs = threadIdx.x >> 5;
    const long long XY = (long long)X * Y;
    const long long off0 = (long long)(ty * 32 + 2 * brs) * X + tx * 64 + 2 * bcp, off1 = off0 + X;
    const int z0 = ch * zlen, z1 = z0 + zlen < Z ? z0 + zlen : Z;
    v2f r0[PFD], r1[PFD];
#pragma unroll
    for (int q = 0; q < PFD; q++) {
        const int z = z0 + q < z1 ? z0 + q : z1 - 1;
        r0[q] = *reinterpret_cast<const v2f *>(a + z * XY + off0); r1[q] = *reinterpret_cast<const v2f *>(a + z * XY + off1);
    }
    for (int z = z0; z < z1; z += PFD) {
#pragma unroll
        for (int q = 0; q < PFD; q++) {
            if (z + q >= z1) break;
            const v2f p0 = r0[q], p1 = r1[q];
            const int zn = z + q + PFD < z1 ? z + q + PFD : z1 - 1;
            r0[q] = *reinterpret_cast<const v2f *>(a + zn * XY + off0); r1[q] = *reinterpret_cast<const v2f *>(a + zn * XY + off1);
            __builtin_nontemporal_store(p0, reinterpret_cast<v2f *>(b + (z + q) * XY + off0));
            __builtin_nontemporal_store(p1, reinterpret_cast<v2f *>(b + (z + q) * XY + off1));
            __builtin_nontemporal_store(p0 + p0, reinterpret_cast<v2f *>(c + (z + q) * XY + off0));
            __builtin_nontemporal_store(p1 + p1, reinterpret_cast<v2f *>(c + (z + q) * XY + off1));
        }
    }
}
// the march with tiles of other shapes at the same area: LX lanes (of 2 floats) per row, 512 threads, 2 rows per thread ->
// tile = 2*LX floats wide by 2*(512/LX) rows: LX 32 = 64 x 32, 64 = 128 x 16, 128 = 256 x 8, 256 = 512 x 4
template <int PFD, int LX>
__global__ __launch_bounds__(1024) void k_tile_w(const float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int X, int Y, int Z, int zlen, int tiles_x, int tiles_y, long long total)
{
    constexpr int TW = 2 * LX;
    const int TH = 2 * ((int)blockDim.x / LX);
    const long long lin = blockIdx.x, per = (total + 7) / 8, w = (lin % 8) * per + lin / 8;
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
        r0[q] = *reinterpret_cast<const v2f *>(a + z * XY + off0); r1[q] = *reinterpret_cast<const v2f *>(a + z * XY + off1);
    }
    for (int z = z0; z < z1; z += PFD) {
#pragma unroll
        for (int q = 0; q < PFD; q++) {
            if (z + q >= z1) break;
            const v2f p0 = r0[q], p1 = r1[q];
            const int zn = z + q + PFD < z1 ? z + q + PFD : z1 - 1;
            r0[q] = *reinterpret_cast<const v2f *>(a + zn * XY + off0); r1[q] = *reinterpret_cast<const v2f *>(a + zn * XY + off1);
            __builtin_nontemporal_store(p0, reinterpret_cast<v2f *>(b + (z + q) * XY + off0));
            __builtin_nontemporal_store(p1, reinterpret_cast<v2f *>(b + (z + q) * XY + off1));
            __builtin_nontemporal_store(p0 + p0, reinterpret_cast<v2f *>(c + (z + q) * XY + off0));
            __builtin_nontemporal_store(p1 + p1, reinterpret_cast<v2f *>(c + (z + q) * XY + off1));
        }
    }
}
// the same march with 16-byte accesses: a thread owns 4 consecutive x of ONE row of the 64 x 32 tile (16 lanes per row)
template <int PFD>
__global__ __launch_bounds__(512) void k_tile16(const float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int X, int Y, int Z, int zlen, int tiles_x, int tiles_y, long long total)
{
    const long long lin = blockIdx.x, per = (total + 7) / 8, w = (lin % 8) * per + lin / 8;
    if (w >= total) return;
    const int tx = (int)(w % tiles_x), ty = (int)((w / tiles_x) % tiles_y), ch = (int)(w / ((long long)tiles_x * tiles_y));
    const long long XY = (long long)X * Y;
    const long long off = (long long)(ty * 32 + (threadIdx.x >> 4)) * X + tx * 64 + 4 * (threadIdx.x & 15);
    const int z0 = ch * zlen, z1 = z0 + zlen < Z ? z0 + zlen : Z;
    v4f r[PFD];
#pragma unroll
    for (int q = 0; q < PFD; q++) r[q] = *reinterpret_cast<const v4f *>(a + (z0 + q < z1 ? z0 + q : z1 - 1) * XY + off);
    for (int z = z0; z < z1; z += PFD) {
#pragma unroll
        for (int q = 0; q < PFD; q++) {
            if (z + q >= z1) break;
            const v4f p = r[q];
            r[q] = *reinterpret_cast<const v4f *>(a + (z + q + PFD < z1 ? z + q + PFD : z1 - 1) * XY + off);
            __builtin_nontemporal_store(p, reinterpret_cast<v4f *>(b + (z + q) * XY + off));
            __builtin_nontemporal_store(p + p, reinterpret_cast<v4f *>(c + (z + q) * XY + off));
        }
    }
}
int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 20;
    const long long n = (long long)N * N * N;
    const long long stag = argc > 3 ? atoll(argv[3]) : 0; /* bytes, a multiple of 16 */
    float *a, *b0, *c0;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b0, n * 4 + 2 * stag + 256)); CK(hipMalloc(&c0, n * 4 + 2 * stag + 256));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b0, 0, n * 4)); CK(hipMemset(c0, 0, n * 4));
    float *b = b0 + stag / 4, *c = c0 + 2 * stag / 4;
    printf("N=%d stagger=%lld bytes  a=%p b=%p c=%p\n", N, stag, (void *)a, (void *)b, (void *)c);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes, auto launch) {
        std::vector<float> ms;
        for (int r = 0; r < reps + 2; r++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); if (r >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const float med = ms[ms.size() / 2];
        printf("%-34s %.3f ms  %.0f GB/s\n", name, med, bytes / med / 1e6);
    };
    const int grid = 256 * 8;
    timeit("copy (1 read + 1 write, float4)", 8.0 * n, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, n / 4); });
    timeit("1 read + 2 writes, float4", 12.0 * n, [&] { hipLaunchKernelGGL(k_1r2w, dim3(grid), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, (v4f *)c, n / 4, 0); });
    timeit("1 read + 2 writes, float4, nt stores", 12.0 * n, [&] { hipLaunchKernelGGL(k_1r2w, dim3(grid), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, (v4f *)c, n / 4, 1); });
    timeit("copy, 4 loads in flight per thread, one-shot grid", 8.0 * n, [&] { hipLaunchKernelGGL(k_copy4, dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, n / 4); });
    timeit("1 read + 2 writes, 4 loads in flight, one-shot", 12.0 * n, [&] { hipLaunchKernelGGL(k_1r2w4, dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, (v4f *)c, n / 4); });
    for (int nch : {2, 4}) {
        const int tiles_x = N / 64, tiles_y = N / 32, zlen = (N + nch - 1) / nch;
        const long long total = (long long)tiles_x * tiles_y * nch, per = (total + 7) / 8;
        char nm[96];
        snprintf(nm, sizeof nm, "tile march 64x32, %d chunks, 1 plane in flight", nch);
        timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(k_tile_pf<1>, dim3((unsigned)(8 * per)), dim3(512), 0, 0, a, b, c, N, N, N, zlen, tiles_x, tiles_y, total); });
        snprintf(nm, sizeof nm, "tile march 64x32, %d chunks, 2 planes in flight", nch);
        timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(k_tile_pf<2>, dim3((unsigned)(8 * per)), dim3(512), 0, 0, a, b, c, N, N, N, zlen, tiles_x, tiles_y, total); });
        snprintf(nm, sizeof nm, "tile march 64x32, %d chunks, 4 planes in flight", nch);
        timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(k_tile_pf<4>, dim3((unsigned)(8 * per)), dim3(512), 0, 0, a, b, c, N, N, N, zlen, tiles_x, tiles_y, total); });
        snprintf(nm, sizeof nm, "tile march 64x32, %d chunks, 16-B accesses, 2 planes in flight", nch);
        timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(k_tile16<2>, dim3((unsigned)(8 * per)), dim3(512), 0, 0, a, b, c, N, N, N, zlen, tiles_x, tiles_y, total); });
        snprintf(nm, sizeof nm, "tile march 64x32, %d chunks, 16-B accesses, 4 planes in flight", nch);
        timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(k_tile16<4>, dim3((unsigned)(8 * per)), dim3(512), 0, 0, a, b, c, N, N, N, zlen, tiles_x, tiles_y, total); });
        snprintf(nm, sizeof nm, "tile march 64x32, %d chunks, 8 planes in flight", nch);
        timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(k_tile_pf<8>, dim3((unsigned)(8 * per)), dim3(512), 0, 0, a, b, c, N, N, N, zlen, tiles_x, tiles_y, total); });
    }
    {
        const int nch = 2, zlen = (N + nch - 1) / nch;
        auto run_w = [&](const char *nm, auto kern, int tw, int th) {
            const int tiles_x = N / tw, tiles_y = N / th;
            const long long total = (long long)tiles_x * tiles_y * nch, per = (total + 7) / 8;
            timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)(8 * per)), dim3(512), 0, 0, a, b, c, N, N, N, zlen, tiles_x, tiles_y, total); });
        };
        run_w("tile shape  64 x 32, 2 chunks, 2 planes in flight", k_tile_w<2, 32>, 64, 32);
        run_w("tile shape 128 x 16, 2 chunks, 2 planes in flight", k_tile_w<2, 64>, 128, 16);
        run_w("tile shape 256 x  8, 2 chunks, 2 planes in flight", k_tile_w<2, 128>, 256, 8);
        run_w("tile shape 512 x  4, 2 chunks, 2 planes in flight", k_tile_w<2, 256>, 512, 4);
        run_w("tile shape  64 x 32 again", k_tile_w<2, 32>, 64, 32);
        auto run_big = [&](const char *nm, auto kern, int tw, int th, int chunks) {   // 1024 threads: twice the area
            const int zl = (N + chunks - 1) / chunks, tiles_x = N / tw, tiles_y = N / th;
            const long long total = (long long)tiles_x * tiles_y * chunks, per = (total + 7) / 8;
            timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)(8 * per)), dim3(1024), 0, 0, a, b, c, N, N, N, zl, tiles_x, tiles_y, total); });
        };
        run_big("1024 thr, tile 128 x 32, 4 chunks (256 WGs), 2 planes", k_tile_w<2, 64>, 128, 32, 4);
        run_big("1024 thr, tile 128 x 32, 4 chunks (256 WGs), 4 planes", k_tile_w<4, 64>, 128, 32, 4);
        run_big("1024 thr, tile 256 x 16, 4 chunks (256 WGs), 2 planes", k_tile_w<2, 128>, 256, 16, 4);
        run_big("1024 thr, tile 512 x  8, 4 chunks (256 WGs), 2 planes", k_tile_w<2, 256>, 512, 8, 4);
        run_big("1024 thr, tile  64 x 64, 4 chunks (256 WGs), 2 planes", k_tile_w<2, 32>, 64, 64, 4);
        run_w("tile shape  64 x 32 once more", k_tile_w<2, 32>, 64, 32);
    }
    struct cfg { int threads, nch, lds; };
    for (cfg c : {cfg{512, 2, 0}, cfg{512, 4, 0}, cfg{512, 4, 90 * 1024}, cfg{512, 8, 90 * 1024}, cfg{1024, 2, 0}, cfg{1024, 4, 0}, cfg{1024, 8, 0}}) {
        const int TYt = 2 * (c.threads / 32);
        const int tiles_x = N / 64, tiles_y = N / TYt, zlen = (N + c.nch - 1) / c.nch;
        const long long total = (long long)tiles_x * tiles_y * c.nch, per = (total + 7) / 8;
        char nm[96]; snprintf(nm, sizeof nm, "tile 64x%d, %d chunks, %lld WGs%s", TYt, c.nch, total, c.lds ? ", 1 WG/CU" : "");
        if (c.lds) hipFuncSetAttribute((const void *)k_tile, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
        timeit(nm, 12.0 * n, [&] { hipLaunchKernelGGL(k_tile, dim3((unsigned)(8 * per)), dim3(c.threads), c.lds, 0, a, b, c0 + 2 * stag / 4, N, N, N, zlen, tiles_x, tiles_y, total); });
    }
    return 0;
}
