// basis_funcs.hpp -- the second dynamics family of the reference:
// GeneralizedLinear<CarBasisFuncs, 7, 2, 25, CarKinematics, 3>
//   PI/generalized_linear.cu:169-245 (computeStateDeriv / computeDynamics / computeKinematics)
//   PI/car_bfs.cuh:44-120            (CarBasisFuncs::basisFuncX, 25 basis functions)
// shared by the device kernel (rollout_bf.hip) and the host replay (nominal trajectory, DDP).
//
// Types follow the source expression by expression.  Where the source divides a float by a double
// literal that is exactly representable in fp32 (10.0, 1200.0, ...), the fp32 quotient is used: a
// single IEEE operation carried out in double and rounded to float equals the fp32 operation
// (p_double >= 2 p_float + 2); that quotient is computed with div_const (exact, see below).  The two sub-expressions that chain double operations --
//   s5/s4 + .45*s6/s4   and   s5/s4 - .35*s6/s4
// -- are evaluated in double as written.  powf(x, 2|3): pow_2 / pow_3 below.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MPPI_HD __host__ __device__ __forceinline__
#else
#define MPPI_HD inline
#endif

namespace mppi {

constexpr int kNumBfs = 25;  // NUM_BFS
constexpr int kBfYThreads = 4;  // BLOCKSIZE_Y of the reference's basis-function build (path_integral_main.cu:73)

// x / c for a constant c with rc = RN(1/c): q0 = x rc, r = fma(-q0, c, x), q = fma(r, rc, q0) is the
// correctly rounded quotient (Markstein); checked against x / c on 60 M random operands for every
// constant used below (scratch check recorded in DESIGN.md) -- 3 operations instead of a division.
MPPI_HD float div_const(float x, float c, float rc)
{
  const float q0 = x * rc;
  const float r = fmaf(-q0, c, x);
  return fmaf(r, rc, q0);
}
#define MPPI_DIVC(x, c) div_const((x), (c), 1.0f / (c))
// the same in double, for the two quotients whose numerator is a genuine double (checked on 100 M operands)
MPPI_HD double div_const_d(double x, double c, double rc)
{
  const double q0 = x * rc;
  const double r = fma(-q0, c, x);
  return fma(r, rc, q0);
}

// powf(x, 2) and powf(x, 3) of the source (car_bfs.cuh:63-64, 90-91, 103, 117-123).  The reference's host code
// (GeneralizedLinear::computeDynamics on the host, generalized_linear.cu:140-167, behind updateState and the
// numerical Jacobian of the DDP, ddp_dynamics.h:71-84) is compiled by g++ and calls the C library's powf;
// the host replays here do the same, so that they evaluate f(z +- h) with the statement of the source (the
// fp32 central differences turn every ulp of f into 1e-4 .. 1e-2 of a Jacobian entry, which 250 Riccati
// steps amplify to tens of percent of a gain).  The device has no powf of that pedigree (CUDA's device
// powf is a different, <= 2-ulp routine); the rollout kernel takes the correctly rounded products.
MPPI_HD float pow_2(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return x * x;
#else
  return powf(x, 2);
#endif
}
MPPI_HD float pow_3(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return (x * x) * x;
#else
  return powf(x, 3);
#endif
}

// The sub-expressions every basis function shares.
struct BasisShared {
  bool big;    // (double)s4 > .1
  float su;    // sinf(u0)
  float A;     // big ? tanf(atanf(s5/s4 + .45*s6/s4) - u0) : tanf(-u0)
  float r54;   // s5/s4
  double B;    // s5/s4 - .35*s6/s4
};

MPPI_HD void basis_shared_common(const float *s, BasisShared &c, float &q)
{
  const float s4 = s[4], s5 = s[5], s6 = s[6];
  c.big = (s4 >= 0.100000001490116119384765625f);  // (double)s4 > .1  <=>  s4 >= 0.1f
  c.r54 = s5 / s4;
  const double d6 = (double)s6, d4 = (double)s4;
  q = (float)((double)c.r54 + 0.45 * d6 / d4);
  c.B = (double)c.r54 - 0.35 * d6 / d4;
}

// as written in the source, with the C library's sinf / atanf / tanf (host replays)
MPPI_HD void basis_shared_libm(const float *s, float u0, BasisShared &c)
{
  float q;
  basis_shared_common(s, c, q);
  c.su = sinf(u0);
  c.A = c.big ? tanf(atanf(q) - u0) : tanf(-u0);
}

// phi[0..24] = basisFuncX(i, s, u) given the shared sub-expressions; s is the full 7-state.
MPPI_HD void basis_funcs_from(const float *s, float u1, const BasisShared &c, float *phi)
{
  const float s3 = s[3], s4 = s[4], s5 = s[5], s6 = s[6];
  const bool big = c.big;
  const float su = c.su, A = c.A, r54 = c.r54;
  const double B = c.B;
  const float A3 = pow_3(A);
  phi[0] = u1;
  phi[1] = MPPI_DIVC(s4, 10.0f);
  phi[2] = MPPI_DIVC(su * A, 1200.0f);
  phi[3] = MPPI_DIVC(su * A * fabsf(A), 1440000.0f);
  phi[4] = MPPI_DIVC(su * A3, 1728000000.0f);
  phi[5] = MPPI_DIVC(s6 * s5, 25.0f);
  phi[6] = MPPI_DIVC(s6, 10.0f);
  phi[7] = MPPI_DIVC(s5, 10.0f);
  phi[8] = su;
  phi[9] = big ? MPPI_DIVC(r54, 40.0f) : 0.0f;
  phi[10] = MPPI_DIVC(A, 1400.0f);
  phi[11] = MPPI_DIVC(A * fabsf(A), 1960000.0f);
  phi[12] = MPPI_DIVC(A3, 2744000000.0f);
  phi[13] = big ? (float)div_const_d(B, 40.0, 1.0 / 40.0) : 0.0f;
  phi[14] = big ? (float)div_const_d(B * fabs(B), 1600.0, 1.0 / 1600.0) : 0.0f;
  const float Bf = (float)B;
  phi[15] = big ? MPPI_DIVC(pow_3(Bf), 64000.0f) : 0.0f;
  phi[16] = MPPI_DIVC(s6 * s4, 50.0f);
  phi[17] = s3;
  phi[18] = s3 * s6;
  phi[19] = MPPI_DIVC(s3 * s4, 3.0f);
  phi[20] = MPPI_DIVC(s3 * s4 * s6, 5.0f);
  phi[21] = MPPI_DIVC(pow_2(s4), 100.0f);
  phi[22] = MPPI_DIVC(pow_3(s4), 1000.0f);
  phi[23] = pow_2(u1);
  phi[24] = pow_3(u1);
}

// phi[0..24] = basisFuncX(i, s, u), host form
MPPI_HD void basis_funcs(const float *s, float u0, float u1, float *phi)
{
  BasisShared c;
  basis_shared_libm(s, u0, c);
  basis_funcs_from(s, u1, c, phi);
}

// s_der[3..6] = W phi.  The reference's y-threads each sum the basis functions i = y, y+4, ...
// (fma-contracted +=) and add their partial sums with atomicAdd, i.e. in no fixed order
// (generalized_linear.cu:226-243); this takes the order y = 0, 1, 2, 3, like the oracle.
MPPI_HD void basis_dynamics(const float *W, const float *phi, float *d)
{
  for (int j = 0; j < 4; j++) {
    float acc = 0.0f;
    for (int y = 0; y < kBfYThreads; y++) {
      float part = 0.0f;
      for (int i = y; i < kNumBfs; i += kBfYThreads) part = fmaf(W[j * kNumBfs + i], phi[i], part);
      acc += part;
    }
    d[j] = acc;
  }
}

}  // namespace mppi
