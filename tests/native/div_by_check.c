/* The three-operation division by 3 and by 6 of mom6_amd/csrc/tracer.hip (div_by<C>): q = RN(x RN(1/c)), r = x - c q exact in a fused
 * multiply-add, RN(q + r RN(1/c)), sign from the numerator -- against the division, bit for bit, on random values of every exponent,
 * on patterns around 1, 2, 4/3 and 8/3 of every binade, and on the subnormals, where it must FAIL only when the quotient is subnormal
 * (the kernel sends every |x| < 2^-1000 through the division).  Usage: div_by_check <millions of random values>; prints
 * "checked N mismatches M" where M counts mismatches with |x| >= 2^-1000 only. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline double divc(double x, double c, double y) { double q = x * y; double r = fma(-c, q, x); return copysign(fma(r, y, q), x); }
static inline uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double frombits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t rng(uint64_t *s) { uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *s = x; }
static long bad = 0, n = 0, bad_tiny = 0;
static void check(double x) {
  if (x != x || isinf(x)) return;
  const double y3 = 1.0 / 3.0, y6 = 1.0 / 6.0;
  const int tiny = fabs(x) < 0x1p-1000;
  n++;
  if (bits(divc(x, 3.0, y3)) != bits(x / 3.0)) { if (tiny) bad_tiny++; else { bad++; if (bad < 5) printf("x / 3: %a\n", x); } }
  if (bits(divc(x, 6.0, y6)) != bits(x / 6.0)) { if (tiny) bad_tiny++; else { bad++; if (bad < 5) printf("x / 6: %a\n", x); } }
}
int main(int argc, char **argv) {
  const long millions = (argc > 1) ? atol(argv[1]) : 20;
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (long it = 0; it < millions * 1000000L; it++) {
    uint64_t u = rng(&s);
    if ((it & 7) != 0) u = (u & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 40 + (rng(&s) % 80)) << 52);   /* mostly moderate exponents */
    check(frombits(u));
  }
  for (int e = 1; e < 2046; e++)
    for (long m = 0; m < 200; m++) {
      const uint64_t sig[6] = { (uint64_t)m, 0xFFFFFFFFFFFFFull - (uint64_t)m, 0x8000000000000ull + (uint64_t)m, 0x8000000000000ull - (uint64_t)m - 1,
                                (0x5555555555555ull + (uint64_t)m) & 0xFFFFFFFFFFFFFull, (0xAAAAAAAAAAAAAull - (uint64_t)m) & 0xFFFFFFFFFFFFFull };
      for (int q = 0; q < 6; q++) for (int sg = 0; sg < 2; sg++) check(frombits(((uint64_t)sg << 63) | ((uint64_t)e << 52) | sig[q]));
    }
  for (uint64_t u = 0; u < 200000; u++) for (int sg = 0; sg < 2; sg++) check(frombits(((uint64_t)sg << 63) | u * 104729ull));   /* zeros, subnormals */
  printf("checked %ld mismatches %ld (with a subnormal quotient, sent through the division by the kernel: %ld)\n", n, bad, bad_tiny);
  return bad != 0;
}
