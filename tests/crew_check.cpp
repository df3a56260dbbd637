// tests/crew_check.cpp -- CPU check of the slab driver's host threads (3d_sift_cuda_amd/csrc/zs_crew.h), built with
// -fsanitize=thread by `make -C 3d_sift_cuda_amd/csrc tsan` and run by tests/test_abi_and_host.py.
// A crew of W workers + the calling thread is stepped through N steps.  In every step a random number n of ranks takes part;
// rank r adds to its own cell what its neighbours' cells held after the step before (plain, non-atomic memory: the crew's
// publish / acknowledge protocol is the only ordering there is, so ThreadSanitizer reports any hole in it), and the calling
// thread checks the cells against a serial replay.  Now and then the calling thread pauses long enough for the workers to fall
// asleep, so that the wake-up path runs too.
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "zs_crew.h"

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 5, N = argc > 2 ? atoi(argv[2]) : 20000;
    const int S = W + 1;
    zs_crew crew;
    crew.spin_limit = 256; /* short: the sleeping path must run often */
    std::vector<int> started((size_t)S, 0);
    crew.start(W, [&](int r) { started[(size_t)r] = 1; });
    std::vector<long long> cell((size_t)S, 1), prev((size_t)S, 1), want((size_t)S, 1), wprev((size_t)S, 1);
    unsigned long long s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 33); };
    long long steps_run = 0;
    for (int i = 0; i < N; i++) {
        const int n = 1 + (int)(rnd() % (unsigned)S);
        prev = cell; /* the step reads `prev` (written here, by the calling thread) and writes `cell` */
        crew.run(n, [&](int r) {
            const long long lo = r > 0 ? prev[(size_t)r - 1] : 0, hi = r + 1 < S ? prev[(size_t)r + 1] : 0;
            cell[(size_t)r] = (prev[(size_t)r] * 3 + lo + 2 * hi + r) % 1000003;
        });
        wprev = want;
        for (int r = 0; r < n; r++) {
            const long long lo = r > 0 ? wprev[(size_t)r - 1] : 0, hi = r + 1 < S ? wprev[(size_t)r + 1] : 0;
            want[(size_t)r] = (wprev[(size_t)r] * 3 + lo + 2 * hi + r) % 1000003;
        }
        for (int r = 0; r < S; r++)
            if (cell[(size_t)r] != want[(size_t)r]) {
                fprintf(stderr, "step %d, rank %d: %lld, expected %lld\n", i, r, cell[(size_t)r], want[(size_t)r]);
                return 1;
            }
        steps_run++;
        if (rnd() % 997 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    crew.stop();
    for (int r = 1; r < S; r++)
        if (!started[(size_t)r]) {
            fprintf(stderr, "worker %d never ran its start hook\n", r);
            return 1;
        }
    /* a crew that was never started runs everything on the calling thread */
    zs_crew lone;
    int hits = 0;
    lone.run(4, [&](int) { hits++; });
    lone.stop();
    if (hits != 4) return 1;
    printf("crew ok: %d workers, %lld steps\n", W, steps_run);
    return 0;
}
