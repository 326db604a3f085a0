// Exhaustive host-side verification of csrc/logadd_fast.h: for EVERY f32 argument of the fast-path
// domains, the fast binary64 evaluation rounded to f32 must equal the correctly rounded result
// (x87 long double expl / log1pl rounded once, the oracle's FCDO_MATH_CR definition) whenever the Ziv
// test calls it safe; also counts how often the test sends a value to the slow path, and checks the
// two shortcuts (exp -> +0 below -104, ln_1p(e) -> e below 2^-24).
//   g++ -O2 -std=c++17 -ffp-contract=off -pthread tools/verify/verify_logadd.cpp -o /tmp/verify_logadd
//   /tmp/verify_logadd [threads] [stride]     stride 1 (default) = exhaustive, ~3 min on 8 cores;
//                                              stride k checks every k-th bit pattern (tests/ uses 499)
// r01 exhaustive run: exp 1 118 568 449 arguments, 517 to the slow path, 0 wrong; ln_1p 201 326 593
// arguments, 397 to the slow path, 0 wrong; both shortcuts exact on all 1.9e9 arguments.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../fast_ctc_decode_amd/csrc/logadd_fast.h"

using namespace fcd;

static float f_of(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t u_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

struct Tally { std::atomic<uint64_t> n{0}, unsafe{0}, bad{0}; };

static uint64_t g_stride = 1;

template <class F>
static void sweep(uint32_t lo, uint32_t hi, int threads, F body) {
    std::vector<std::thread> th;
    const uint64_t span = (uint64_t)hi - lo + 1, step = (span + threads - 1) / threads;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([=] {
            const uint64_t a = lo + t * step, b = a + step < (uint64_t)hi + 1 ? a + step : (uint64_t)hi + 1;
            for (uint64_t u = a; u < b; u += g_stride) body((uint32_t)u);
        });
    for (auto &x : th) x.join();
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : (int)std::thread::hardware_concurrency();
    if (argc > 2) g_stride = strtoull(argv[2], nullptr, 10);
    if (g_stride < 1) g_stride = 1;
    int rc = 0;
    {   // exp: every f32 in [-86, -0]  (bit patterns 0x80000000 .. bits(-86))
        Tally t;
        sweep(0x80000000u, u_of(kExpFastMin), threads, [&](uint32_t u) {
            const float x = f_of(u);
            const double y = exp_fast((double)x);
            t.n++;
            if (round_to_f32_unsafe(y)) { t.unsafe++; return; }
            const float want = (float)expl((long double)x);
            if (u_of((float)y) != u_of(want) || round_to_f32_as_f64(y) != (double)(float)y) {
                if (t.bad++ < 5) printf("exp MISMATCH x=%a fast=%a want=%a\n", x, (float)y, want);
            }
        });
        printf("exp   : %llu arguments, %llu sent to the slow path (%.2e), %llu wrong\n",
               (unsigned long long)t.n, (unsigned long long)t.unsafe, (double)t.unsafe / t.n, (unsigned long long)t.bad);
        rc |= t.bad != 0;
    }
    {   // ln_1p: every f32 in [2^-126, 1]
        Tally t;
        sweep(u_of(kLog1pFastMin), u_of(1.0f), threads, [&](uint32_t u) {
            const float e = f_of(u);
            const double y = log1p_fast((double)e);
            t.n++;
            if (round_to_f32_unsafe(y)) { t.unsafe++; return; }
            const float want = (float)log1pl((long double)e);
            if (u_of((float)y) != u_of(want)) {
                if (t.bad++ < 5) printf("log1p MISMATCH e=%a fast=%a want=%a\n", e, (float)y, want);
            }
        });
        printf("ln_1p : %llu arguments, %llu sent to the slow path (%.2e), %llu wrong\n",
               (unsigned long long)t.n, (unsigned long long)t.unsafe, (double)t.unsafe / t.n, (unsigned long long)t.bad);
        rc |= t.bad != 0;
    }
    {   // shortcut: ln_1p(e) == e for every f32 in [+0, 2^-24)
        Tally t;
        sweep(0u, u_of(kLog1pIdentityBelow) - 1, threads, [&](uint32_t u) {
            const float e = f_of(u);
            t.n++;
            if (u_of((float)log1pl((long double)e)) != u) t.bad++;
        });
        printf("ln_1p(e) == e below 2^-24: %llu arguments, %llu wrong\n", (unsigned long long)t.n, (unsigned long long)t.bad);
        rc |= t.bad != 0;
    }
    {   // shortcut: exp(x) == +0 for every finite f32 below -104
        Tally t;
        sweep(u_of(kExpZeroBelow) + 1, 0xFF7FFFFFu, threads, [&](uint32_t u) {
            const float x = f_of(u);
            t.n++;
            if (u_of((float)expl((long double)x)) != 0u) t.bad++;
        });
        printf("exp(x) == +0 below -104: %llu arguments, %llu wrong\n", (unsigned long long)t.n, (unsigned long long)t.bad);
        rc |= t.bad != 0;
    }
    printf(rc ? "FAILED\n" : "all fast paths verified\n");
    return rc;
}
