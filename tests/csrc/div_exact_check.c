/* div_exact (ir_sgmcmc_amd/csrc/common.h) against the IEEE division it replaces: quotient estimate with the correctly
 * rounded reciprocal, exact remainder (FMA), one correction.  Integer divisors n - 1 = 1 .. 2048, 20 000 random dividends
 * each over 60 binades.  Exit code 1 on any mismatch.  Test infrastructure (tests/test_host_logic.py). */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(void) {
    long bad = 0, n = 0;
    for (int d = 1; d <= 2048; ++d) {
        volatile float b = (float)d;
        volatile float rb = 1.0f / b;
        for (int k = 0; k < 20000; ++k) {
            uint32_t bits = (uint32_t)rnd();
            // exponent range 2^-40 .. 2^20, random sign and mantissa
            uint32_t e = 87 + (rnd() % 61);
            bits = (bits & 0x807FFFFFu) | (e << 23);
            float a; memcpy(&a, &bits, 4);
            volatile float q = a * rb;
            volatile float r = fmaf(-q, b, a);
            volatile float q2 = fmaf(r, rb, q);
            volatile float ref = a / b;
            if (q2 != ref) { if (bad < 5) printf("a=%a b=%d got %a want %a\n", a, d, q2, ref); ++bad; }
            ++n;
        }
    }
    printf("checked %ld, mismatches %ld\n", n, bad);
    return bad != 0;
}
