// rollout_bf.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) with the reference's second
// dynamics family, GeneralizedLinear<CarBasisFuncs,7,2,25,CarKinematics,3>
// (PI/generalized_linear.cu:169-245, PI/car_bfs.cuh:44-120), SURVEY 8f row f3.
//
// One lane per rollout: a step is 25 scalar basis functions (one sincos, two f64 divides and a few
// fp32 divides shared by all of them) and a 4 x 25 matrix-vector product -- 100 MACs, nothing a matrix
// instruction could help with -- followed by the same cost evaluation as the network kernels.  W
// (400 B) is staged into LDS and read as broadcasts.  Noise comes from the stand-alone generator.
#include "basis_funcs.hpp"
#include "mppi_kernels.hpp"

namespace mppi {

constexpr int kBfLanes = 64;

// Device form of the shared sub-expressions.  tan(atan(q) - u0) is evaluated through
// tan(a - b) = (tan a - tan b) / (1 + tan a tan b) with tan(atan q) = q, and sin u0 / tan u0 come from
// one sincos_fast: no atanf, no tanf, no large-argument reduction on the recurrence.  The reference's
// own device code composes CUDA's sinf / atanf / tanf (2-4 ulp each); this form stays within the same
// few ulp of the exact value (tests/test_basis_funcs.py: <= 2e-5 of the derivative's scale against the
// literal restatement).
__device__ __forceinline__ void basis_shared_fast(const float *s, float u0, BasisShared &c)
{
  float q, sn, cs;
  basis_shared_common(s, c, q);
  sincos_fast(u0, sn, cs);
  const float t = sn / cs;
  c.su = sn;
  c.A = c.big ? (q - t) / fmaf(q, t, 1.0f) : -t;
}

// computeStateDeriv: kinematics with the yaw rate always negated (generalized_linear.cu:212-217)
__device__ __forceinline__ void bf_state_deriv(const float *W_s, const float *s, float u0, float u1, float cpsi,
                                               float spsi, float *sd)
{
  sd[0] = fmaf(cpsi, s[4], -(spsi * s[5]));
  sd[1] = fmaf(spsi, s[4], cpsi * s[5]);
  sd[2] = -s[6];
  float phi[kNumBfs];
  BasisShared c;
  basis_shared_fast(s, u0, c);
  basis_funcs_from(s, u1, c, phi);
  basis_dynamics(W_s, phi, sd + 3);
}

__global__ __launch_bounds__(kBfLanes) void rollout_bf_kernel(const RolloutArgs a)
{
  __shared__ float W_s[4 * kNumBfs];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4 * kNumBfs; i += kBfLanes) W_s[i] = a.wpack[i];
  __syncthreads();
  const int k = blockIdx.x * kBfLanes + lane;
  if (k >= a.K) return;  // K % 64 == 0: never splits a wave

  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = a.state[i];
  int crash = 0;
  float J = 0.0f;
  const int K = a.K, T = a.T;
  float2 *const noise = reinterpret_cast<float2 *>(a.noise);
  const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
  const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
  const bool pure_noise_k = (k >= a.k99);  // :141

  float2 e_next = noise[(size_t)k];
  for (int t = 0; t < T; t++) {
    const float2 e = e_next;
    if (t + 1 < T) e_next = noise[(size_t)(t + 1) * K + k];
    const float2 Ut = Useq[t];
    float du0, du1, u0, u1;
    if (noise_free_k || t < a.opt_delay) {
      du0 = 0.0f; du1 = 0.0f; u0 = Ut.x; u1 = Ut.y;
    } else {
      du0 = e.x * a.nu[0];
      du1 = e.y * a.nu[1];
      u0 = pure_noise_k ? du0 : Ut.x + du0;
      u1 = pure_noise_k ? du1 : Ut.y + du1;
    }
    noise[(size_t)t * K + k] = make_float2(u0, u1);  // before the clamp (Q3)
    u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
    u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
    float spsi, cpsi;
    sincos_fast(s[2], spsi, cpsi);
    float tf = 0.0f, tb = 0.0f;
    if (t > 0) {
      if (a.cost.affine) track_fetch<true>(a.cost, s, cpsi, spsi, tf, tb);
      else track_fetch<false>(a.cost, s, cpsi, spsi, tf, tb);
    }
    float sd[kStateDim];
    bf_state_deriv(W_s, s, u0, u1, cpsi, spsi, sd);
    if (t > 0) {
      const float c = a.cost.need_control_cost
                          ? cost_finish<true>(a.cost, a.nu, s[4], s[5], tf, tb, u0, u1, du0, du1, crash)
                          : cost_finish<false>(a.cost, a.nu, s[4], s[5], tf, tb, u0, u1, du0, du1, crash);
      J = running_mean(J, c, t, a.inv_t[t]);
    }
#pragma unroll
    for (int i = 0; i < kStateDim; i++) s[i] = fmaf(sd[i], a.dt, s[i]);
    crash |= (int)(fabsf(s[3]) >= kRollCrash);
  }
  a.costs[k] = J + 0.0f;
}

// test entry (mppi_debug_dynamics): state derivative of n independent (state, control) pairs
__global__ __launch_bounds__(kBfLanes) void dynamics_bf_kernel(const float *W, const float *states,
                                                               const float *controls, float *ders, int n)
{
  __shared__ float W_s[4 * kNumBfs];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4 * kNumBfs; i += kBfLanes) W_s[i] = W[i];
  __syncthreads();
  const int idx = blockIdx.x * kBfLanes + lane;
  if (idx >= n) return;
  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = states[idx * kStateDim + i];
  float spsi, cpsi;
  sincos_fast(s[2], spsi, cpsi);
  float sd[kStateDim];
  bf_state_deriv(W_s, s, controls[idx * 2], controls[idx * 2 + 1], cpsi, spsi, sd);
#pragma unroll
  for (int i = 0; i < kStateDim; i++) ders[idx * kStateDim + i] = sd[i];
}

hipError_t launch_rollout_bf(const RolloutArgs &a, hipStream_t stream)
{
  hipLaunchKernelGGL(rollout_bf_kernel, dim3(a.K / kBfLanes), dim3(kBfLanes), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_dynamics_bf(const float *W, const float *states, const float *controls, float *ders, int n,
                              hipStream_t stream)
{
  hipLaunchKernelGGL(dynamics_bf_kernel, dim3((n + kBfLanes - 1) / kBfLanes), dim3(kBfLanes), 0, stream, W, states,
                     controls, ders, n);
  return hipGetLastError();
}

}  // namespace mppi
