/* verify_const_div.c -- offline proof by exhaustion for the 3-instruction division by a small
 * integer constant used in the kernels (rc_kernels.hip, div_by_nk):
 *     y  = RN(1/c)            (host, once)
 *     q0 = RN(a*y)
 *     r  = RN(fma(-c, q0, a)) (exact remainder)
 *     q  = RN(fma(r, y, q0))
 * For every c in 2..31 and EVERY positive normal binary32 a in [2^-100, 2^100) this checks
 * q == RN(a/c) (IEEE division).  By symmetry of all operations the result also holds for
 * negative a.  Output: one line per c, "c ok" or "c FAIL first_a".
 * Build: gcc -O2 -mfma -o verify_const_div verify_const_div.c -lm   (needs a CPU with FMA)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
  for (int c = 2; c <= 31; c++) {
    const float cf = (float)c, y = 1.0f / cf;
    uint32_t lo, hi, fail = 0, bits;
    float t = ldexpf(1.0f, -100); memcpy(&lo, &t, 4);
    t = ldexpf(1.0f, 100); memcpy(&hi, &t, 4);
    for (bits = lo; bits < hi; bits++) {
      float a; memcpy(&a, &bits, 4);
      float q0 = a * y;
      float r = fmaf(-cf, q0, a);
      float q = fmaf(r, y, q0);
      if (q != a / cf) { fail = bits; break; }
    }
    if (fail) printf("%d FAIL 0x%08x\n", c, fail); else printf("%d ok\n", c);
    fflush(stdout);
  }
  return 0;
}
