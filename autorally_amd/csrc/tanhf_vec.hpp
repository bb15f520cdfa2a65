// tanhf_vec.hpp -- tanhf of eight floats per call, BIT FOR BIT the tanhf of the C library the reference's host
// code calls (MPPI_NNET_NONLINEARITY = tanh on Eigen floats: neural_net_model.cu:35,213-228; glibc's float
// tanh is fdlibm's s_tanhf.c on top of s_expm1f.c, glibc 2.35 sysdeps/ieee754/flt-32/).
//
// Why: the host replays of the network (computeNominalTraj, the DDP forward pass) spend most of their time in
// 64 scalar tanhf calls per step, ~20 ns each; a merely accurate substitute changes state_solution_ in the last
// digit and the feedback gains by 1e-3 through the Riccati recursion, so the substitute has to
// return libm's own bits.  This is that algorithm -- the same IEEE single-precision operations in the same order --
// with every branch turned into a lane select.  Proven equal to the installed libm on all 2^32 inputs by
// tools/tanhf_exhaustive (run by tests/test_host_math.py over a stride of the inputs; NaN payloads aside), and
// checked again at library load (tanhf_vec_selfcheck): if the installed tanhf ever differs, the host replays
// fall back to calling libm.
//
// expm1f paths tanhf reaches (a = 2|x| in [2, 44) or a = -2|x| in (-2, 0)):
//   |a| < 2^-25: a;  |a| <= ln2/2: k = 0;  -1.5 ln2 < a < -ln2/2: k = -1;  a <= -1.5 ln2: k in {-2, -3};
//   a >= 2: k in [3, 63], forms k < 23, 23 <= k <= 56, k > 56.
#pragma once
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace mppi {

// compile with -ffp-contract=off (both build.py and the host Makefile do): no multiply-add may be fused here
static inline __m256 tanhf8(__m256 x)
{
  const __m256 one = _mm256_set1_ps(1.0f), two = _mm256_set1_ps(2.0f), half = _mm256_set1_ps(0.5f);
  const __m256i abs_mask = _mm256_set1_epi32(0x7fffffff);
  const __m256i jx = _mm256_castps_si256(x);
  const __m256i ix = _mm256_and_si256(jx, abs_mask);
  const __m256 ax = _mm256_castsi256_ps(ix);  // fabsf(x)
  // tanhf: |x| >= 1 -> expm1f(2|x|), else expm1f(-2|x|)
  const __m256 ge1 = _mm256_castsi256_ps(_mm256_cmpgt_epi32(ix, _mm256_set1_epi32(0x3f7fffff)));
  const __m256 twoax = _mm256_mul_ps(two, ax);
  const __m256 a = _mm256_blendv_ps(_mm256_xor_ps(twoax, _mm256_set1_ps(-0.0f)), twoax, ge1);  // -two*|x| == -(two*|x|)

  // ---- expm1f(a) ----
  const __m256 ln2_hi = _mm256_castsi256_ps(_mm256_set1_epi32(0x3f317180));
  const __m256 ln2_lo = _mm256_castsi256_ps(_mm256_set1_epi32(0x3717f7d1));
  const __m256 invln2 = _mm256_castsi256_ps(_mm256_set1_epi32(0x3fb8aa3b));
  const __m256 Q1 = _mm256_castsi256_ps(_mm256_set1_epi32((int)0xbd088889));
  const __m256 Q2 = _mm256_castsi256_ps(_mm256_set1_epi32(0x3ad00d01));
  const __m256 Q3 = _mm256_castsi256_ps(_mm256_set1_epi32((int)0xb8a670cd));
  const __m256 Q4 = _mm256_castsi256_ps(_mm256_set1_epi32(0x36867e54));
  const __m256 Q5 = _mm256_castsi256_ps(_mm256_set1_epi32((int)0xb457edbb));
  const __m256i hx = _mm256_and_si256(_mm256_castps_si256(a), abs_mask);
  const __m256 neg = _mm256_castsi256_ps(_mm256_srai_epi32(_mm256_castps_si256(a), 31));  // xsb != 0
  // argument reduction
  const __m256 red = _mm256_castsi256_ps(_mm256_cmpgt_epi32(hx, _mm256_set1_epi32(0x3eb17218)));  // |a| > 0.5 ln2
  const __m256 near = _mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_set1_epi32(0x3F851592), hx));  // |a| < 1.5 ln2
  const __m256 sh = _mm256_blendv_ps(half, _mm256_set1_ps(-0.5f), neg);
  __m256i k = _mm256_cvttps_epi32(_mm256_add_ps(_mm256_mul_ps(invln2, a), sh));  // k = invln2*x + (+-0.5), truncated
  // the special case |a| < 1.5 ln2: k = -1 (a < 0; a > 0 never comes here from tanhf: a >= 2), same hi / lo formulas
  const __m256 km1 = _mm256_and_ps(_mm256_and_ps(red, near), neg);
  k = _mm256_castps_si256(_mm256_blendv_ps(_mm256_castsi256_ps(k), _mm256_castsi256_ps(_mm256_set1_epi32(-1)), km1));
  k = _mm256_and_si256(k, _mm256_castps_si256(red));  // no reduction: k = 0
  const __m256 tk = _mm256_cvtepi32_ps(k);
  const __m256 hi = _mm256_sub_ps(a, _mm256_mul_ps(tk, ln2_hi));  // t*ln2_hi is exact
  const __m256 lo = _mm256_mul_ps(tk, ln2_lo);
  const __m256 xr = _mm256_sub_ps(hi, lo);                       // k = 0: hi = a, lo = 0 -> a
  const __m256 c = _mm256_sub_ps(_mm256_sub_ps(hi, xr), lo);     // k = 0: 0
  // primary range
  const __m256 hfx = _mm256_mul_ps(half, xr);
  const __m256 hxs = _mm256_mul_ps(xr, hfx);
  __m256 r1 = _mm256_add_ps(Q4, _mm256_mul_ps(hxs, Q5));
  r1 = _mm256_add_ps(Q3, _mm256_mul_ps(hxs, r1));
  r1 = _mm256_add_ps(Q2, _mm256_mul_ps(hxs, r1));
  r1 = _mm256_add_ps(Q1, _mm256_mul_ps(hxs, r1));
  r1 = _mm256_add_ps(one, _mm256_mul_ps(hxs, r1));
  const __m256 t3 = _mm256_sub_ps(_mm256_set1_ps(3.0f), _mm256_mul_ps(r1, hfx));
  const __m256 e = _mm256_mul_ps(hxs, _mm256_div_ps(_mm256_sub_ps(r1, t3), _mm256_sub_ps(_mm256_set1_ps(6.0f), _mm256_mul_ps(xr, t3))));
  // k == 0: x - (x*e - hxs)
  const __m256 r_k0 = _mm256_sub_ps(xr, _mm256_sub_ps(_mm256_mul_ps(xr, e), hxs));
  // k != 0
  __m256 e2 = _mm256_sub_ps(_mm256_mul_ps(xr, _mm256_sub_ps(e, c)), c);
  e2 = _mm256_sub_ps(e2, hxs);
  const __m256 r_km1 = _mm256_sub_ps(_mm256_mul_ps(half, _mm256_sub_ps(xr, e2)), half);  // 0.5*(x-e) - 0.5
  const __m256i kshift = _mm256_slli_epi32(k, 23);
  // k <= -2 || k > 56: y = one - (e - x); exponent += k; y - one
  __m256 y_a = _mm256_sub_ps(one, _mm256_sub_ps(e2, xr));
  y_a = _mm256_castsi256_ps(_mm256_add_epi32(_mm256_castps_si256(y_a), kshift));
  const __m256 r_far = _mm256_sub_ps(y_a, one);
  // 0 < k < 23: t = 1 - 2^-k; y = t - (e - x); exponent += k
  const __m256i kpos = _mm256_max_epi32(k, _mm256_setzero_si256());
  const __m256 t_b = _mm256_castsi256_ps(_mm256_sub_epi32(_mm256_set1_epi32(0x3f800000), _mm256_srlv_epi32(_mm256_set1_epi32(0x1000000), kpos)));
  __m256 y_b = _mm256_sub_ps(t_b, _mm256_sub_ps(e2, xr));
  y_b = _mm256_castsi256_ps(_mm256_add_epi32(_mm256_castps_si256(y_b), kshift));
  // 23 <= k <= 56: t = 2^-k; y = x - (e + t); y += one; exponent += k
  const __m256 t_c = _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_sub_epi32(_mm256_set1_epi32(0x7f), k), 23));
  __m256 y_c = _mm256_add_ps(_mm256_sub_ps(xr, _mm256_add_ps(e2, t_c)), one);
  y_c = _mm256_castsi256_ps(_mm256_add_epi32(_mm256_castps_si256(y_c), kshift));
  const __m256 is_k0 = _mm256_castsi256_ps(_mm256_cmpeq_epi32(k, _mm256_setzero_si256()));
  const __m256 is_km1 = _mm256_castsi256_ps(_mm256_cmpeq_epi32(k, _mm256_set1_epi32(-1)));
  const __m256 is_far = _mm256_or_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_set1_epi32(-1), k)),
                                     _mm256_castsi256_ps(_mm256_cmpgt_epi32(k, _mm256_set1_epi32(56))));
  const __m256 is_b = _mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_set1_epi32(23), k));  // k < 23 (k > 0 by exclusion)
  __m256 em1 = _mm256_blendv_ps(y_c, y_b, is_b);
  em1 = _mm256_blendv_ps(em1, r_far, is_far);
  em1 = _mm256_blendv_ps(em1, r_km1, is_km1);
  em1 = _mm256_blendv_ps(em1, r_k0, is_k0);
  // |a| < 2^-25: expm1f returns its argument
  const __m256 tiny_a = _mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_set1_epi32(0x33000000), hx));
  em1 = _mm256_blendv_ps(em1, a, tiny_a);

  // ---- tanhf ----
  const __m256 den = _mm256_add_ps(em1, two);
  const __m256 z_ge1 = _mm256_sub_ps(one, _mm256_div_ps(two, den));                      // one - two/(t+two)
  const __m256 z_lt1 = _mm256_div_ps(_mm256_xor_ps(em1, _mm256_set1_ps(-0.0f)), den);    // -t/(t+two)
  __m256 z = _mm256_blendv_ps(z_lt1, z_ge1, ge1);
  // |x| >= 22 (and +-inf): one - tiny == 1
  const __m256 big = _mm256_castsi256_ps(_mm256_cmpgt_epi32(ix, _mm256_set1_epi32(0x41afffff)));
  z = _mm256_blendv_ps(z, one, big);
  // sign: (jx >= 0) ? z : -z
  z = _mm256_xor_ps(z, _mm256_and_ps(x, _mm256_set1_ps(-0.0f)));
  // |x| < 2^-55: x*(one+x); zero: x (the same expression gives +-0)
  const __m256 small = _mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_set1_epi32(0x24000000), ix));
  z = _mm256_blendv_ps(z, _mm256_mul_ps(x, _mm256_add_ps(one, x)), small);
  // NaN: one/x +- one is a NaN; x + x is one too
  const __m256 nan = _mm256_cmp_ps(x, x, _CMP_UNORD_Q);
  z = _mm256_blendv_ps(z, _mm256_add_ps(x, x), nan);
  return z;
}

// n values in place; n need not be a multiple of 8 (the tail goes through a padded register)
static inline void tanhf_vec(float *v, int n)
{
  int i = 0;
  for (; i + 8 <= n; i += 8) _mm256_storeu_ps(v + i, tanhf8(_mm256_loadu_ps(v + i)));
  if (i < n) {
    float tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(tmp, v + i, sizeof(float) * (size_t)(n - i));
    _mm256_storeu_ps(tmp, tanhf8(_mm256_loadu_ps(tmp)));
    memcpy(v + i, tmp, sizeof(float) * (size_t)(n - i));
  }
}

// Is tanhf8 the installed libm's tanhf?  A few thousand inputs over every branch (the exhaustive comparison is
// tools/tanhf_exhaustive).  The host replays call libm instead when this fails.
static inline bool tanhf_vec_selfcheck()
{
  uint32_t s = 0x9E3779B9u;
  for (int it = 0; it < 4096; it += 8) {
    float in[8], out[8];
    for (int q = 0; q < 8; q++) {
      s = s * 1664525u + 1013904223u;
      uint32_t b = s;
      if ((it & 24) != 24) {  // three quarters: |x| spread over [2^-30, 32), where the branches are
        const uint32_t ex = 97u + ((s >> 8) % 35u);
        b = (s & 0x807fffffu) | (ex << 23);
      }
      memcpy(&in[q], &b, 4);
    }
    _mm256_storeu_ps(out, tanhf8(_mm256_loadu_ps(in)));
    for (int q = 0; q < 8; q++) {
      const float r = tanhf(in[q]);
      if (memcmp(&r, &out[q], 4) != 0 && !(r != r && out[q] != out[q])) return false;
    }
  }
  return true;
}

}  // namespace mppi
