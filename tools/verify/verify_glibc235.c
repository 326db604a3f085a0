/* tools/verify/verify_glibc235.c -- csrc/glibc235_math.h against the C library this program links, on EVERY binary32
 * argument (2^32 each for expf, logf, log1pf; NaN results compare as equal).  On glibc 2.35 / x86-64 with FMA: 0
 * differences.  Build and run (about three minutes on one core):
 *     g++ -O2 -ffp-contract=off -fno-fast-math -I fast_ctc_decode_amd/csrc tools/verify/verify_glibc235.c -o /tmp/vg -lm && /tmp/vg
 */
#include <gnu/libc-version.h>
#include <stdio.h>

#include "glibc235_math.h"

int main(void) {
    unsigned long long bad[3] = {0, 0, 0};
    for (unsigned long long u = 0; u <= 0xFFFFFFFFull; ++u) {
        const float x = fcd::g235::asfloat((uint32_t)u);
        const float want[3] = {expf(x), logf(x), log1pf(x)};
        const float got[3] = {fcd::g235::expf235(x), fcd::g235::logf235(x), fcd::g235::log1pf235(x)};
        for (int k = 0; k < 3; ++k)
            if (fcd::g235::asuint(want[k]) != fcd::g235::asuint(got[k]) && !(want[k] != want[k] && got[k] != got[k])) {
                if (bad[k] < 4) printf("%s(%a): libm %a, restated %a\n", k == 0 ? "expf" : k == 1 ? "logf" : "log1pf", x, want[k], got[k]);
                ++bad[k];
            }
    }
    printf("libm: glibc %s; 4294967296 arguments each; differences: expf %llu, logf %llu, log1pf %llu\n",
           gnu_get_libc_version(), bad[0], bad[1], bad[2]);
    return bad[0] || bad[1] || bad[2];
}
