// Host check of exact_div::div_by_known (csrc/exact_div.h) against the IEEE division it replaces.
// Build: g++ -O2 -ffp-contract=off -o exact_div_test exact_div_test.cpp ; prints "OK <cases>" and exits 0.
#include "../../moving_object_detector_amd/csrc/exact_div.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

static uint64_t bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
static bool same(double a, double b) { return bits(a) == bits(b) || (a != a && b != b); }

int main(int argc, char **argv) {
  const long per_d = argc > 1 ? atol(argv[1]) : 400000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> uni(-1.0, 1.0);
  const float specials[] = {0.0f, -0.0f, 1.0f, -1.0f, 1.17549435e-38f, 1.4e-45f, -1.4e-45f, 3.40282347e38f, -3.40282347e38f,
                            INFINITY, -INFINITY, NAN, 0.1f, 0.3f, 1e-20f, 1e20f};
  long cases = 0, bad = 0;
  for (int k = 0; k < 400; k++) {
    double d;
    if (k < 8) { const double fixed[] = {0.1, 1.0 / 30.0, 0.05, 1.0, 0.033333333, 0.1000000001, 1.0 / 3.0, 2.5}; d = fixed[k]; }
    else if (k < 40) d = ldexp(1.0 + 0.5 * (uni(rng) + 1.0) * 0.999999, (int)(uni(rng) * 199));
    else d = ldexp(1.0 + (double)(rng() & ((1ull << 52) - 1)) / 4503599627370496.0, (int)(uni(rng) * 30)) * ((rng() & 1) ? 1 : -1);
    if (!exact_div::reciprocal_usable(d)) continue;
    const double rd = 1.0 / d;
    for (long i = 0; i < per_d; i++) {
      float nf;
      const uint64_t r = rng();
      if (i < (long)(sizeof(specials) / sizeof(float))) nf = specials[i];
      else if (r & 1) { uint32_t u = (uint32_t)(r >> 16); memcpy(&nf, &u, 4); }     // any F32 bit pattern
      else nf = (float)(uni(rng) * ldexp(1.0, (int)(uni(rng) * 20)));               // ordinary magnitudes
      const double n = (double)nf;
      const double want = n / d, got = exact_div::div_by_known(n, std::isnormal(nf) || std::fpclassify(nf) == FP_SUBNORMAL, d, rd);
      cases++;
      if (!same(want, got)) { if (bad++ < 10) printf("MISMATCH n=%a d=%a want=%a got=%a\n", n, d, want, got); }
    }
  }
  if (bad) { printf("FAILED %ld of %ld\n", bad, cases); return 1; }
  printf("OK %ld\n", cases);
  return 0;
}
