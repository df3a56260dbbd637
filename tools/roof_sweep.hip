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
    // workgroup order: plain against XCD-aware, on the blur's own shape
    for (int xcd : {0, 1}) run(k_march<2, 2, 2, 2>, 2, 2, 2, 2, 32, 512, 2, xcd);
    for (int xcd : {0, 1}) run(k_march<2, 2, 2, 2>, 2, 2, 2, 2, 64, 512, 2, xcd);
    return 0;
}
