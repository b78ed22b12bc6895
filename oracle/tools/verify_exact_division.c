/* Exhaustive check of the two division replacements used by the fast transform kernels
 * (TEST INFRASTRUCTURE; the kernels carry the same formulas).
 *
 *   byte / 255.0f  ==  fmaf(x, hi, x * lo)          for all 256 byte values
 *   x / q          ==  fmaf(x, rhi, x * rlo)        for every integer divisor q in 1..255 and
 *                                                   every float mantissa of one binade, both
 *                                                   signs (scaling by powers of two is exact,
 *                                                   so one binade stands for all normal x)
 * with hi+lo and rhi+rlo the float pair nearest to the exact reciprocal.  Prints the
 * number of mismatches; exit status 0 iff there are none.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -fopenmp verify_exact_division.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(int argc, char** argv) {
    const unsigned step = argc > 1 ? (unsigned)atoi(argv[1]) : 1; /* mantissa stride, 1 = exhaustive */
    long bad = 0;
    {
        const double inv = 1.0 / 255.0;
        const float hi = (float)inv, lo = (float)(inv - (double)hi);
        for (int u = 0; u < 256; ++u) {
            const float x = (float)u;
            if (fmaf(x, hi, x * lo) != x / 255.0f) ++bad;
        }
        printf("inv255 hi=%a lo=%a\n", hi, lo);
    }
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1)
    for (int q = 1; q <= 255; ++q) {
        const double inv = 1.0 / (double)q;
        const float rhi = (float)inv, rlo = (float)(inv - (double)rhi), qf = (float)q;
        for (uint32_t m = 0; m < (1u << 23); m += step) {
            const uint32_t bits = 0x43800000u | m; /* [256, 512) */
            float x;
            memcpy(&x, &bits, 4);
            if (fmaf(x, rhi, x * rlo) != x / qf) ++bad;
            x = -x;
            if (fmaf(x, rhi, x * rlo) != x / qf) ++bad;
        }
    }
    printf("mismatches=%ld\n", bad);
    return bad == 0 ? 0 : 1;
}
