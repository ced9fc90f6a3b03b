// TEST TOOL.  Exhaustive check (all 2^31 magnitudes, ~1 min; `divcheck STRIDE` samples every STRIDE-th
// and all patterns near the boundaries) of the constant-division sequence the gate's producer wave
// uses (div_const_fast in rfid_kernels.hpp): q = fma(fma(-q1, c, x), rc, q1) with q1 = x*rc, rc = RN(1/c)  ==  RN(x / c) ?
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
static inline float fast_div(float x, float c, float rc) {
  float q1 = x * rc;
  float r = __builtin_fmaf(-q1, c, x);
  return __builtin_fmaf(r, rc, q1);
}
int main(int argc, char **argv) {
  const uint64_t stride = (argc > 1) ? strtoull(argv[1], 0, 10) : 1;
  const float cs[2] = {100.0f, 48.0f};
  for (int ci = 0; ci < 2; ++ci) {
    const float c = cs[ci];
    const float rc = 1.0f / c;
    uint64_t bad = 0; uint32_t lo_bad_max = 0, hi_bad_min = 0x7f800000u;
    for (uint64_t u = 0; u < 0x7f800000ull; u += ((u >= 0x0d000000ull && u < 0x0e000000ull) || u >= 0x7f700000ull) ? 1 : stride) {   // finite, non-negative (sign symmetric)
      uint32_t b = (uint32_t)u; float x; memcpy(&x, &b, 4);
      volatile float ref = x / c;
      float q = fast_div(x, c, rc);
      uint32_t bq, br; float rf = ref; memcpy(&bq, &q, 4); memcpy(&br, &rf, 4);
      if (bq != br) {
        bad++;
        if (b < 0x3f800000u) { if (b > lo_bad_max) lo_bad_max = b; } else { if (b < hi_bad_min) hi_bad_min = b; }
      }
    }
    printf("c=%g rc=%a mismatches=%llu largest_bad_bits=0x%08x smallest_large_bad_bits=0x%08x\n",
           c, rc, (unsigned long long)bad, lo_bad_max, hi_bad_min);
  }
  return 0;
}

