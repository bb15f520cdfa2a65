// noise_device.hpp -- device functions of the noise spec: MRG32k3a step and
// jump, and the Box-Muller transform written only in IEEE basic operations.  Shared by the
// stand-alone generator kernel (noise_mrg32k3a.hip) and the control wavefront of the quad rollout
// kernel (rollout_mfma.hip); both must produce the same bits as a CPU statement of the spec.
#pragma once

#include "mppi_device.hpp"

namespace mppi {

constexpr uint64_t kM1 = 4294967087ULL, kM2 = 4294944443ULL;
constexpr uint32_t kC1 = 209u, kC2 = 22853u;  // 2^32 mod m1, 2^32 mod m2
constexpr uint64_t kA12 = 1403580ULL, kA13N = 810728ULL, kA21 = 527612ULL, kA23N = 1370589ULL;

__device__ __forceinline__ uint32_t fold_m1(uint64_t x)
{  // x < 2^63 -> x mod m1
  x = (x >> 32) * kC1 + (x & 0xffffffffULL);
  x = (x >> 32) * kC1 + (x & 0xffffffffULL);
  if (x >= kM1) x -= kM1;
  return (uint32_t)x;
}
__device__ __forceinline__ uint32_t fold_m2(uint64_t x)
{
  x = (x >> 32) * kC2 + (x & 0xffffffffULL);
  x = (x >> 32) * kC2 + (x & 0xffffffffULL);
  x = (x >> 32) * kC2 + (x & 0xffffffffULL);
  if (x >= kM2) x -= kM2;
  return (uint32_t)x;
}

struct Mrg {
  uint32_t s10, s11, s12, s20, s21, s22;
};

__device__ __forceinline__ uint32_t mrg_next_z(Mrg &g)
{
  // p1 = (a12*s11 - a13n*s10) mod m1 ; p2 = (a21*s22 - a23n*s20) mod m2  (kept positive)
  const uint32_t p1 = fold_m1(kA12 * g.s11 + kA13N * (kM1 - g.s10));
  g.s10 = g.s11; g.s11 = g.s12; g.s12 = p1;
  const uint32_t p2 = fold_m2(kA21 * g.s22 + kA23N * (kM2 - g.s20));
  g.s20 = g.s21; g.s21 = g.s22; g.s22 = p2;
  uint64_t z = (p1 >= p2) ? (uint64_t)(p1 - p2) : (uint64_t)p1 + kM1 - p2;
  if (z == 0) z = kM1;
  return (uint32_t)z;
}

__device__ __forceinline__ void mat3_vec_m1(const uint32_t *A, uint32_t &a, uint32_t &b, uint32_t &c)
{
  uint32_t r[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const uint64_t acc = (uint64_t)fold_m1((uint64_t)A[3 * i] * a) +
                         fold_m1((uint64_t)A[3 * i + 1] * b) + fold_m1((uint64_t)A[3 * i + 2] * c);
    r[i] = fold_m1(acc);
  }
  a = r[0]; b = r[1]; c = r[2];
}
__device__ __forceinline__ void mat3_vec_m2(const uint32_t *A, uint32_t &a, uint32_t &b, uint32_t &c)
{
  uint32_t r[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const uint64_t acc = (uint64_t)fold_m2((uint64_t)A[3 * i] * a) +
                         fold_m2((uint64_t)A[3 * i + 1] * b) + fold_m2((uint64_t)A[3 * i + 2] * c);
    r[i] = fold_m2(acc);
  }
  a = r[0]; b = r[1]; c = r[2];
}

// log(x), x normal positive: fdlibm e_logf evaluated exactly as written in the spec.
__device__ __forceinline__ float spec_logf(float x)
{
  const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
  const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
  uint32_t ix = __float_as_uint(x);
  ix += 0x3f800000u - 0x3f3504f3u;
  const int k = (int)(ix >> 23) - 0x7f;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  x = __uint_as_float(ix);
  const float f = x - 1.0f;
  const float s = f / (2.0f + f);
  const float z = s * s;
  const float w = z * z;
  const float t1 = w * (Lg2 + w * Lg4);
  const float t2 = z * (Lg1 + w * Lg3);
  const float R = t2 + t1;
  const float hfsq = (0.5f * f) * f;
  const float dk = (float)k;
  return (((s * (hfsq + R) + dk * ln2_lo) - hfsq) + f) + dk * ln2_hi;
}

__device__ __forceinline__ void spec_sincos2pi(float v, float &sn, float &cs)
{
  const float S1 = -0.16666667163f, S2 = 0.0083333291113f, S3 = -0.00019839334709f, S4 = 2.7183114e-06f;
  const float C0 = -0.5f, C1 = 0.041666623205f, C2 = -0.0013886763947f, C3 = 2.4390449e-05f;
  const float x = v - 0.5f;
  const float kq = rintf(x * 4.0f);
  const float y = x - kq * 0.25f;
  const float a = y * 6.2831853071795864769f;
  const float a2 = a * a;
  const float ps = S1 + a2 * (S2 + a2 * (S3 + a2 * S4));
  const float sin_a = a + (a * a2) * ps;
  const float pc = C0 + a2 * (C1 + a2 * (C2 + a2 * C3));
  const float cos_a = 1.0f + a2 * pc;
  const int q = ((int)kq) & 3;
  float s2, c2;
  if (q == 0) { s2 = sin_a; c2 = cos_a; }
  else if (q == 1) { s2 = cos_a; c2 = -sin_a; }
  else if (q == 2) { s2 = -sin_a; c2 = -cos_a; }
  else { s2 = -cos_a; c2 = sin_a; }
  sn = -s2;
  cs = -c2;
}

// One timestep's pair of uniforms: two generator steps.
__device__ __forceinline__ float2 uniform_pair(Mrg &g)
{
  const float u1 = (float)mrg_next_z(g) * 0x1p-32f;
  const float u2 = (float)mrg_next_z(g) * 0x1p-32f;
  return make_float2(u1, u2);
}

// Box-Muller on one pair of uniforms.
__device__ __forceinline__ float2 box_muller(float2 u)
{
  const float r = sqrtf(-2.0f * spec_logf(u.x));
  float sn, cs;
  spec_sincos2pi(u.y, sn, cs);
  return make_float2(r * sn, r * cs);
}

// One timestep's pair of N(0,1) draws: two generator steps + Box-Muller.
__device__ __forceinline__ float2 noise_pair(Mrg &g)
{
  return box_muller(uniform_pair(g));
}

}  // namespace mppi
