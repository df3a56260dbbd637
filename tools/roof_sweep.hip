// tools/roof_sweep.hip -- development aid (GPU box): which ACCESS SHAPE of a z march streams best, with no arithmetic.
// Round 5: the fused blur sits at 0.91 of the zero-arithmetic march of its own tiles (bench.py: roofline.frac_of_ceiling) and that
// march at 0.67 of the HBM peak, against 0.77 for a flat stream.  This sweep asks what the march's shape costs and whether another
// shape of the same march would raise the ceiling: tile width (row segment per workgroup), threads per workgroup, z chunks
// (= concurrent read planes), planes in flight per thread, 8- or 16-byte accesses, one or two store streams, workgroup order.
//   usage: roof_sweep [N=512] [reps=7]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// a thread owns VW consecutive x of RPT rows (RPT rows apart by 1) of the tile; LX lanes cover a tile row of LX*VW floats
template <int PFD, int VW, int RPT, int NW>
__global__ __launch_bounds__(1024) void k_march(const float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int X, int Y, int Z, int zlen, int LX,
                                                int tiles_x, int tiles_y, long long total, int xcd_order)
{
    typedef float vec __attribute__((ext_vector_type(VW)));
    const int TW = VW * LX, TH = RPT * ((int)blockDim.x / LX);
    const long long lin = blockIdx.x, per = (total + 7) / 8;
    const long long w = xcd_order ? (lin % 8) * per + lin / 8 : lin;
    if (w >= total) return;
    const int tx = (int)(w % tiles_x), ty = (int)((w / tiles_x) % tiles_y), ch = (int)(w / ((long long)tiles_x * tiles_y));
    const int bcp = threadIdx.x % LX, brs = threadIdx.x / LX;
    const long long XY = (long long)X * Y;
    long long off[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) off[r] = (long long)(ty * TH + RPT * brs + r) * X + tx * TW + VW * bcp;
    const int z0 = ch * zlen, z1 = z0 + zlen < Z ? z0 + zlen : Z;
    vec ring[PFD][RPT];
#pragma unroll
    for (int q = 0; q < PFD; q++) {
        const int z = z0 + q < z1 ? z0 + q : z1 - 1;
#pragma unroll
        for (int r = 0; r < RPT; r++) ring[q][r] = *reinterpret_cast<const vec *>(a + z * XY + off[r]);
    }
    for (int z = z0; z < z1; z += PFD) {
#pragma unroll
        for (int q = 0; q < PFD; q++) {
            if (z + q >= z1) break;
            vec p[RPT];
            const int zn = z + q + PFD < z1 ? z + q + PFD : z1 - 1;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                p[r] = ring[q][r];
                ring[q][r] = *reinterpret_cast<const vec *>(a + zn * XY + off[r]);
            }
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                __builtin_nontemporal_store(p[r], reinterpret_cast<vec *>(b + (z + q) * XY + off[r]));
                if (NW > 1) __builtin_nontemporal_store(p[r] + p[r], reinterpret_cast<vec *>(c + (z + q) * XY + off[r]));
            }
        }
    }
}

// The access shape of a FUSED PAIR of filters (round-4 review item 3), with no arithmetic: a workgroup loads a TW x TH region of
// the input but owns only the inner part (HX columns and HY rows less on every side: the second filter's halo, which the first
// filter has to produce on the ring); it stores the inner part to TWO arrays.  Tiles therefore overlap: pitch (TW - 2 HX) x (TH - 2 HY).
template <int PFD>
__global__ __launch_bounds__(512) void k_march_pair(const float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int X, int Y, int Z, int zlen, int LX,
                                                    int HX, int HY, int tiles_x, int tiles_y, long long total)
{
    typedef float vec __attribute__((ext_vector_type(2)));
    const int TW = 2 * LX, TH = 2 * ((int)blockDim.x / LX), PW = TW - 2 * HX, PH = TH - 2 * HY;
    const long long lin = blockIdx.x, per = (total + 7) / 8;
    const long long w = (lin % 8) * per + lin / 8;
    if (w >= total) return;
    const int tx = (int)(w % tiles_x), ty = (int)((w / tiles_x) % tiles_y), ch = (int)(w / ((long long)tiles_x * tiles_y));
    const int bcp = threadIdx.x % LX, brs = threadIdx.x / LX;
    const long long XY = (long long)X * Y;
    const int gx = tx * PW - HX + 2 * bcp;
    long long off[2];
    bool ld[2], st[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int ry = 2 * brs + r, gy = ty * PH - HY + ry;
        ld[r] = gx >= 0 && gx + 1 < X && gy >= 0 && gy < Y;
        st[r] = ld[r] && 2 * bcp >= HX && 2 * bcp < TW - HX && ry >= HY && ry < TH - HY;
        off[r] = (long long)gy * X + gx;
    }
    const int z0 = ch * zlen, z1 = z0 + zlen < Z ? z0 + zlen : Z;
    vec ring[PFD][2];
#pragma unroll
    for (int q = 0; q < PFD; q++) {
        const int z = z0 + q < z1 ? z0 + q : z1 - 1;
#pragma unroll
        for (int r = 0; r < 2; r++) ring[q][r] = ld[r] ? *reinterpret_cast<const vec *>(a + z * XY + off[r]) : vec(0.0f);
    }
    for (int z = z0; z < z1; z += PFD) {
#pragma unroll
        for (int q = 0; q < PFD; q++) {
            if (z + q >= z1) break;
            vec p[2];
            const int zn = z + q + PFD < z1 ? z + q + PFD : z1 - 1;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                p[r] = ring[q][r];
                ring[q][r] = ld[r] ? *reinterpret_cast<const vec *>(a + zn * XY + off[r]) : vec(0.0f);
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (st[r]) {
                    __builtin_nontemporal_store(p[r], reinterpret_cast<vec *>(b + (z + q) * XY + off[r]));
                    __builtin_nontemporal_store(p[r] + p[r], reinterpret_cast<vec *>(c + (z + q) * XY + off[r]));
                }
        }
    }
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 7;
    const long long n = (long long)N * N * N;
    float *a, *b, *c;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) {
        std::vector<float> ms;
        for (int r = 0; r < reps + 2; r++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); if (r >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return ms[ms.size() / 2];
    };
    printf("# N=%d: zero-arithmetic z march, median of %d launches; GB/s = (4 + 4*stores) B/voxel / time; frac of 8000 GB/s\n", N, reps);
    printf("# %-7s %-8s %-7s %-6s %-6s %-4s %-6s %-5s %-6s %8s %8s %6s\n", "tile", "threads", "rows/t", "vecB", "chunks", "PFD", "stores", "xcd", "WGs", "ms", "GB/s", "frac");
    struct shape { int lx, vw, rpt, threads; };
    auto run = [&](auto kern, int PFD, int VW, int RPT, int NW, int LX, int threads, int chunks, int xcd) {
        const int tw = VW * LX, th = RPT * (threads / LX);
        if (threads % LX || tw > N || th > N || N % tw || N % th || th < 1) return;
        const int tiles_x = N / tw, tiles_y = N / th, zlen = (N + chunks - 1) / chunks;
        const long long total = (long long)tiles_x * tiles_y * chunks, per = (total + 7) / 8;
        const unsigned grid = (unsigned)(xcd ? 8 * per : total);
        const float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, a, b, c, N, N, N, zlen, LX, tiles_x, tiles_y, total, xcd); });
        const double gbs = (4.0 + 4.0 * NW) * n / ms / 1e6;
        printf("%4dx%-3d %-8d %-7d %-6d %-6d %-4d %-6d %-5d %-6lld %8.4f %8.0f %6.3f\n", tw, th, threads, RPT, 4 * VW, chunks, PFD, NW, xcd, total, ms, gbs, gbs / 8000.0);
        fflush(stdout);
    };
    for (int NW : {2, 1})
        for (int threads : {256, 512, 1024})
            for (int tw : {64, 128, 256, 512})
                for (int chunks : {1, 2, 4, 8}) {
                    // 8-byte accesses, two rows per thread (the blur's own mapping)
                    if (NW == 2) {
                        run(k_march<2, 2, 2, 2>, 2, 2, 2, 2, tw / 2, threads, chunks, 1);
                        run(k_march<4, 2, 2, 2>, 4, 2, 2, 2, tw / 2, threads, chunks, 1);
                        run(k_march<4, 4, 1, 2>, 4, 4, 1, 2, tw / 4, threads, chunks, 1);
                        run(k_march<8, 4, 1, 2>, 8, 4, 1, 2, tw / 4, threads, chunks, 1);
                        run(k_march<4, 4, 2, 2>, 4, 4, 2, 2, tw / 4, threads, chunks, 1);
                    } else {
                        run(k_march<2, 2, 2, 1>, 2, 2, 2, 1, tw / 2, threads, chunks, 1);
                        run(k_march<4, 4, 1, 1>, 4, 4, 1, 1, tw / 4, threads, chunks, 1);
                        run(k_march<4, 4, 2, 1>, 4, 4, 2, 1, tw / 4, threads, chunks, 1);
                    }
                }
    // the fused pair's shape: load TW x TH, own and store the inner (TW - 2 HX) x (TH - 2 HY) to two arrays (12 B/voxel of useful traffic)
    printf("# fused-pair shape (load tile, halo, inner tile, chunks, WGs): ms, GB/s of the 12 B/voxel it is credited, frac; compare: two level-only launches = 2 x the 1-store march\n");
    auto run_pair = [&](int LX, int threads, int HX, int HY, int chunks, int PFD) {
        const int tw = 2 * LX, th = 2 * (threads / LX), pw = tw - 2 * HX, ph = th - 2 * HY;
        const int tiles_x = (N + pw - 1) / pw, tiles_y = (N + ph - 1) / ph, zlen = (N + chunks - 1) / chunks;
        const long long total = (long long)tiles_x * tiles_y * chunks, per = (total + 7) / 8;
        const float ms = timeit([&] {
            if (PFD == 2) hipLaunchKernelGGL(k_march_pair<2>, dim3((unsigned)(8 * per)), dim3(threads), 0, 0, a, b, c, N, N, N, zlen, LX, HX, HY, tiles_x, tiles_y, total);
            else hipLaunchKernelGGL(k_march_pair<4>, dim3((unsigned)(8 * per)), dim3(threads), 0, 0, a, b, c, N, N, N, zlen, LX, HX, HY, tiles_x, tiles_y, total);
        });
        const double gbs = 12.0 * n / ms / 1e6;
        printf("pair %3dx%-3d halo %dx%d inner %3dx%-3d chunks %d PFD %d WGs %-5lld %8.4f ms %6.0f GB/s %6.3f\n", tw, th, HX, HY, pw, ph, chunks, PFD, total, ms, gbs, gbs / 8000.0);
        fflush(stdout);
    };
    for (int PFD : {2, 4})
        for (int chunks : {1, 2}) {
            run_pair(32, 512, 8, 3, chunks, PFD);   // 64 x 32 loaded, 48 x 26 owned (16-byte aligned segments)
            run_pair(32, 512, 4, 3, chunks, PFD);   // 64 x 32 loaded, 56 x 26 owned (8-byte aligned segments)
            run_pair(64, 512, 8, 3, chunks, PFD);   // 128 x 16 loaded, 112 x 10 owned
            run_pair(64, 512, 4, 3, chunks, PFD);   // 128 x 16 loaded, 120 x 10 owned
            run_pair(128, 512, 4, 3, chunks, PFD);  // 256 x 8 loaded, 248 x 2 owned (the y halo eats the tile)
        }
    // workgroup order: plain against XCD-aware, on the blur's own shape
    for (int xcd : {0, 1}) run(k_march<2, 2, 2, 2>, 2, 2, 2, 2, 32, 512, 2, xcd);
    for (int xcd : {0, 1}) run(k_march<2, 2, 2, 2>, 2, 2, 2, 2, 64, 512, 2, xcd);
    return 0;
}
