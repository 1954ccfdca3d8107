/* TEST INFRASTRUCTURE (tests/test_cpu_properties.py): exhaustive host check of the 3-operation quotient the decoder
 * epilogues use for the MRF mean xs / num_kernels (wetts_amd/csrc/common.h: div_small_const):
 *     q = RN(v * c), r = fma(-d, q, v), result = fma(r, c, q),   c = RN(1 / d)
 * against the IEEE division v / d for ALL 2^32 float bit patterns.  Prints the number of finite inputs whose result
 * differs in any bit other than the sign of a zero, and the number of sign-of-zero differences (expected: 0 and 1,
 * v = -0.0f).   usage: div_small_const_check <d> */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
  const float d = argc > 1 ? (float)atof(argv[1]) : 3.0f;
  const float c = 1.0f / d;
  uint64_t bad = 0, zsign = 0;
#pragma omp parallel for reduction(+ : bad, zsign) schedule(static, 1 << 20)
  for (uint64_t i = 0; i < (1ull << 32); ++i) {
    uint32_t u = (uint32_t)i, w, g;
    float v;
    memcpy(&v, &u, 4);
    if (!isfinite(v)) continue;
    const float want = v / d;
    const float q = v * c;
    const float r = fmaf(-d, q, v);
    const float got = fmaf(r, c, q);
    memcpy(&w, &want, 4);
    memcpy(&g, &got, 4);
    if (w == g) continue;
    if (want == 0.0f && got == 0.0f) zsign++; else bad++;
  }
  printf("%llu %llu\n", (unsigned long long)bad, (unsigned long long)zsign);
  return 0;
}
