// Test harness (not product): append_g (kpop_text.h; std::to_chars) against printf("%.*g"), the format the reference
// prints every number in (lib/Matrix.ml:684-690, Printf "%.*g").  argv[1] = seed, argv[2] = values.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>

#include "../../kpop_amd/host/kpop_text.h"

using namespace kpop_host;

int main(int argc, char **argv) {
  if (argc < 3) return 64;
  std::mt19937_64 rng((uint64_t)atoll(argv[1]));
  const size_t n = (size_t)atoll(argv[2]);
  char want[600];
  std::string got;
  size_t bad = 0;
  const double fixed[] = {0.0, -0.0, 1.0, -1.0, 0.1, 1e-5, 9.9999999999999995e-5, 1e15, 1e16, 123456789012345.0, 1234567890123456.0,
                          999999999999999.5, 0.000123456789012345678, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, INFINITY,
                          -INFINITY, NAN, -NAN, 0.5, 2.5, 1e21, 1e22, 1e23, 4.35, 0.3, 2.675};
  for (size_t i = 0; i < n + sizeof(fixed) / sizeof(fixed[0]); ++i) {
    double x;
    const uint64_t bits = rng();
    if (i >= n) x = fixed[i - n];
    else switch (i % 6) {
      case 0: memcpy(&x, &bits, 8); break;                                         // any bit pattern (NaNs, subnormals, ...)
      case 1: x = (double)(int64_t)(bits % 2000001) / 1000.0 - 1000.0; break;      // short decimals
      case 2: x = ldexp((double)(bits >> 11), -53); break;                          // [0, 1)
      case 3: x = (double)(bits % 100000) * 0.5; break;                             // integers and halves
      case 4: x = ldexp(1.0 + (double)(bits >> 12) * 0x1p-52, (int)(bits % 120) - 60); break;
      default: x = (double)(bits % 1000000007ull) * 1e-9 * 123456.789; break;       // distances, z scores
    }
    for (int precision : {15, 17, 6, 1, 0, 20}) {
      const int len = snprintf(want, sizeof(want), "%.*g", precision, x);
      got.clear();
      append_g(got, x, precision);
      if ((size_t)len != got.size() || memcmp(want, got.data(), got.size())) {
        if (++bad < 10) printf("%a at precision %d: printf '%s', append_g '%s'\n", x, precision, want, got.c_str());
      }
    }
  }
  printf("%zu values, %zu differ\n", n, bad);
  return bad ? 2 : 0;
}
