// mppi_device.hpp -- argument blocks and device helpers shared by the gfx950 kernels.
//
// All kernels are compiled with -ffp-contract=off: every fused multiply-add is an
// explicit fmaf()/MFMA placed where the reference's nvcc build contracts one
// (SURVEY 8c "fidelity rules"); everything else rounds exactly as written.
// Reference citations are relative to /root/reference/autorally_control/,
//   PI/ = include/autorally_control/path_integral/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mppi {

constexpr int kStateDim = 7;
constexpr int kControlDim = 2;
constexpr int kNetIn = 6;   // [roll, u_x, u_y, yaw_mder, steering, throttle]
constexpr int kNetOut = 4;  // d/dt [roll, u_x, u_y, yaw_mder]
constexpr int kRolloutsPerWave = 16;  // N dimension of v_mfma_f32_16x16x4_f32

// Scalars of MPPICosts::CostParams (PI/costs.cuh:67-85) in the form the kernels consume.
struct CostArgs {
  float desired_speed, speed_coeff, track_coeff, max_slip_ang, slip_penalty, track_slop;
  float crash_coeff, steering_coeff, throttle_coeff, boundary_threshold;
  float crash_cost_discounted;  // (float)((1.0 - (double)discount) * (double)crash_coeff), costs.cu:402 (Q6)
  int l1_cost;
  float r_c1[3], r_c2[3], trs[3];
  int affine;     // r_c1.z == 0 && r_c2.z == 0 && trs.z == 1  => w == 1 exactly, u/w == u
  int need_control_cost;  // steering_coeff != 0 || throttle_coeff != 0 || nu not finite/non-zero
  int map_w, map_h;
  const float *map;  // channel 0 plane, [H][W]
};

struct RolloutArgs {
  float state[kStateDim];
  const float *U;   // [T][2]
  float *noise;     // [T][K][2]: in N(0,1), out applied unclamped control (Q3)
  float *costs;     // [K]
  const float *wpack;  // MFMA-ordered weights (see pack_mfma_weights) or packed theta (VALU kernel)
  int K, T, opt_delay, k99;
  float nu[2], u_lo[2], u_hi[2], dt;
  int negate_yaw_der;
  CostArgs cost;
};

// Thresholds for the reference's float-vs-double-literal comparisons, as floats:
//   (double)x > 1.57   <=>  x >= kRollCrash   (costs.cu:302)
//   (double)x > 0.001  <=>  x >= kMinSpeed    (costs.cu:340)
//   (double)x > 1e12   <=>  x >= kCostCapGt   (costs.cu:405); replacement value (float)1e12
__device__ constexpr float kRollCrash = 1.57000005245208740234375f;     // nextafter((float)1.57 < 1.57 ? ...)
__device__ constexpr float kMinSpeed = 0.001000000047497451305389404296875f;
__device__ constexpr float kCostCapGt = 1000000061440.0f;
__device__ constexpr float kCostCap = 999999995904.0f;

__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
  // enforceConstraints, neural_net_model.cu:311-323 (NaN passes through unchanged)
  if (v < lo) v = lo;
  else if (v > hi) v = hi;
  return v;
}

// tanh for the hidden layers (MPPI_NNET_NONLINEARITY, neural_net_model.cu:35).
// 1 - 2/(exp(2x)+1) on the transcendental unit: |err| <= ~2e-7 absolute over the real line,
// saturates correctly (+-1) for large |x|, keeps NaN.
__device__ __forceinline__ float tanh_fast(float x)
{
  const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);  // exp(2x)
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return fmaf(-2.0f, r, 1.0f);
}

__device__ __forceinline__ float texel_x(const CostArgs &c, float x, float y)
{
  // coorTransform (costs.cu:351-357) + point/clamp/normalised tex2D (costs.cu:128-154)
  float u = fmaf(c.r_c1[0], x, c.r_c2[0] * y) + c.trs[0];
  float v = fmaf(c.r_c1[1], x, c.r_c2[1] * y) + c.trs[1];
  if (!c.affine) {
    const float w = fmaf(c.r_c1[2], x, c.r_c2[2] * y) + c.trs[2];
    u = u / w;
    v = v / w;
  }
  float fi = floorf(u * (float)c.map_w);
  float fj = floorf(v * (float)c.map_h);
  if (!(fi >= 0.0f)) fi = 0.0f;
  if (!(fj >= 0.0f)) fj = 0.0f;
  fi = fminf(fi, (float)(c.map_w - 1));
  fj = fminf(fj, (float)(c.map_h - 1));
  const int i = (int)fi, j = (int)fj;
  return c.map[(size_t)j * (size_t)c.map_w + (size_t)i];
}

// MPPICosts::computeCost (costs.cu:396-409).  cpsi/spsi = cos/sin of s[2] (the reference uses
// __cosf/__sinf here, Q7; this build reuses the precise values already needed by the kinematics).
__device__ __forceinline__ float compute_cost(const CostArgs &c, const float nu[2], const float *s,
                                              float cpsi, float spsi, float u0, float u1, float du0,
                                              float du1, int &crash)
{
  float control_cost = 0.0f;
  if (c.need_control_cost) {  // getControlCost :307-313
    control_cost += c.steering_coeff * du0 * (u0 - du0) / (nu[0] * nu[0]);
    control_cost += c.throttle_coeff * du1 * (u1 - du1) / (nu[1] * nu[1]);
  }
  // getTrackCost :359-393
  const float xf = fmaf(0.5f, cpsi, s[0]), yf = fmaf(0.5f, spsi, s[1]);
  const float xb = fmaf(-0.5f, cpsi, s[0]), yb = fmaf(-0.5f, spsi, s[1]);
  const float tf = texel_x(c, xf, yf);
  const float tb = texel_x(c, xb, yb);
  float track_cost = (fabsf(tf) + fabsf(tb)) * 0.5f;  // == (float)((double)(..)/2.0)
  track_cost = (fabsf(track_cost) < c.track_slop) ? 0.0f : c.track_coeff * track_cost;
  if (tf >= c.boundary_threshold || tb >= c.boundary_threshold) crash = 1;
  // getSpeedCost :315-326
  const float err = s[4] - c.desired_speed;
  const float speed_cost = c.speed_coeff * (c.l1_cost ? fabsf(err) : err * err);
  // (1.0 - discount) * getCrashCost :402, :328-335
  const float crash_cost = (crash > 0) ? c.crash_cost_discounted : 0.0f;
  // getStabilizingCost :337-349
  float stabilizing_cost = 0.0f;
  if (fabsf(s[4]) >= kMinSpeed) {
    const float slip = -atanf(s[5] / fabsf(s[4]));
    stabilizing_cost = c.slip_penalty * (slip * slip);
    if (fabsf(slip) > c.max_slip_ang) stabilizing_cost += c.crash_coeff;
  }
  float cost = control_cost + speed_cost + crash_cost + track_cost + stabilizing_cost;
  if (cost >= kCostCapGt || cost != cost) cost = kCostCap;
  return cost;
}

// running_cost += (cost - running_cost)/(1.0*i)  in double (mppi_controller.cu:165, Q5)
__device__ __forceinline__ float running_mean(float J, float c, int t)
{
  return (float)((double)J + (double)(c - J) / (double)t);
}

}  // namespace mppi
