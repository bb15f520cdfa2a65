// tools/host/tanhf_exhaustive.cpp: csrc/tanhf_vec.hpp against the installed libm's tanhf.
//   tanhf_exhaustive [stride]   stride 1 (default): all 2^32 bit patterns (~25 s on 8 cores); stride s: every s-th
// Prints the number of inputs whose results differ in any bit (NaN results compare equal to NaN); exit code 1 if any.
// g++ -O2 -mavx2 -mfma -ffp-contract=off -fopenmp tools/host/tanhf_exhaustive.cpp -o tanhf_exhaustive
#include <cstdio>
#include <cstdlib>
#include "../../autorally_amd/csrc/tanhf_vec.hpp"

int main(int argc, char **argv)
{
  const long long stride = argc > 1 ? atoll(argv[1]) : 1;
  unsigned long long bad = 0, n = 0;
  const long long blocks = ((1LL << 32) + 8 * stride - 1) / (8 * stride);
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
  for (long long blk = 0; blk < blocks; blk++) {
    float in[8], out[8];
    for (int q = 0; q < 8; q++) {
      const uint32_t b = (uint32_t)((blk * 8 + q) * stride);
      memcpy(&in[q], &b, 4);
    }
    _mm256_storeu_ps(out, mppi::tanhf8(_mm256_loadu_ps(in)));
    for (int q = 0; q < 8; q++) {
      const float r = tanhf(in[q]);
      n++;
      if (memcmp(&r, &out[q], 4) != 0 && !(r != r && out[q] != out[q])) {
        bad++;
        if (bad < 5) {
          uint32_t bi, br, bo;
          memcpy(&bi, &in[q], 4); memcpy(&br, &r, 4); memcpy(&bo, &out[q], 4);
          fprintf(stderr, "x=%08x (%g): libm %08x vec %08x\n", bi, in[q], br, bo);
        }
      }
    }
  }
  printf("tanhf8 vs libm tanhf: %llu inputs, %llu differ; selfcheck %s\n", n, bad, mppi::tanhf_vec_selfcheck() ? "ok" : "FAILED");
  return bad != 0;
}
