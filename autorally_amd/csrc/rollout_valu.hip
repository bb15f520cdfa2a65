// rollout_valu.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) on the vector ALU, for ANY
// layer list (the reference fixes the net by template arguments, path_integral_main.cu:69).
//
// One lane per rollout, one wavefront per workgroup.  The packed parameter blob
// [W1|b1|W2|b2|...] (neural_net_model.cu:120-141) is staged once into LDS (the reference reads it
// from global memory on every use, Q1); activations ping-pong through a lane-major LDS tile
// act[i][lane] (bank = lane, conflict free), weights are LDS broadcasts.  This is the fallback
// for shapes the MFMA kernel does not cover and the "vector-ALU" arm of the SURVEY cfg-4 A/B.
#include "mppi_kernels.hpp"

namespace mppi {

constexpr int kValuLanes = 64;

struct NetDev {
  int n_layers;
  int layers[8];
  int max_width;
  int num_params;
};

// d[0..3] = NN(in[0..5]); cur/nxt: LDS tiles [max_width][64]
__device__ __forceinline__ void nn_forward_valu(const NetDev &net, const float *theta_s, float *cur,
                                                float *nxt, int lane, const float (&in)[kNetIn],
                                                float (&d)[4])
{
#pragma unroll
  for (int i = 0; i < kNetIn; i++) cur[i * kValuLanes + lane] = in[i];
  int off = 0;
  for (int l = 0; l + 1 < net.n_layers; l++) {
    const int nin = net.layers[l], nout = net.layers[l + 1];
    const float *W = theta_s + off;
    const float *b = W + nout * nin;
    const bool hidden = (l < net.n_layers - 2);
    for (int jn = 0; jn < nout; jn++) {
      float tmp = 0.0f;
      const float *Wr = W + jn * nin;
      for (int kk = 0; kk < nin; kk++) tmp = fmaf(Wr[kk], cur[kk * kValuLanes + lane], tmp);
      // same formula as the MFMA kernel (tanh_bias) so that the two arms stay bit-identical
      tmp = hidden ? tanh_bias(tmp, b[jn] * kTanhScale) : tmp + b[jn];
      nxt[jn * kValuLanes + lane] = tmp;
    }
    off += nout * nin + nout;
    float *t = cur; cur = nxt; nxt = t;
  }
#pragma unroll
  for (int i = 0; i < kNetOut; i++) d[i] = cur[i * kValuLanes + lane];
}

__global__ __launch_bounds__(kValuLanes) void rollout_valu_kernel(const RolloutArgs a, const NetDev net)
{
  extern __shared__ float lds[];
  float *theta_s = lds;
  float *act0 = lds + ((net.num_params + 3) & ~3);
  float *act1 = act0 + net.max_width * kValuLanes;
  const int lane = threadIdx.x;
  for (int i = lane; i < net.num_params; i += kValuLanes) theta_s[i] = a.wpack[i];
  __syncthreads();
  const int k = blockIdx.x * kValuLanes + lane;
  if (k >= a.K) return;  // K % 64 == 0: never splits a wave

  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = a.state[i];
  int crash = 0;
  float J = 0.0f;
  const int K = a.K, T = a.T;
  float2 *const noise = reinterpret_cast<float2 *>(a.noise);
  const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
  const bool noise_free_k = (k == 0);
  const bool pure_noise_k = (k >= a.k99);

  for (int t = 0; t < T; t++) {
    const float2 e = noise[(size_t)t * K + k];
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
    noise[(size_t)t * K + k] = make_float2(u0, u1);
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
    sd[0] = fmaf(cpsi, s[4], -(spsi * s[5]));
    sd[1] = fmaf(spsi, s[4], cpsi * s[5]);
    sd[2] = a.negate_yaw_der ? -s[6] : s[6];
    const float in[kNetIn] = {s[3], s[4], s[5], s[6], u0, u1};
    float d[4];
    nn_forward_valu(net, theta_s, act0, act1, lane, in, d);
    sd[3] = d[0]; sd[4] = d[1]; sd[5] = d[2]; sd[6] = d[3];
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

__global__ __launch_bounds__(kValuLanes) void dynamics_valu_kernel(const NetDev net, const float *theta,
                                                                   const float *states,
                                                                   const float *controls, float *ders,
                                                                   int n, int negate_yaw_der)
{
  extern __shared__ float lds[];
  float *theta_s = lds;
  float *act0 = lds + ((net.num_params + 3) & ~3);
  float *act1 = act0 + net.max_width * kValuLanes;
  const int lane = threadIdx.x;
  for (int i = lane; i < net.num_params; i += kValuLanes) theta_s[i] = theta[i];
  __syncthreads();
  const int idx = blockIdx.x * kValuLanes + lane;
  const int src = idx < n ? idx : n - 1;
  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = states[src * kStateDim + i];
  const float in[kNetIn] = {s[3], s[4], s[5], s[6], controls[src * 2], controls[src * 2 + 1]};
  float spsi, cpsi;
  sincos_fast(s[2], spsi, cpsi);
  float d[4];
  nn_forward_valu(net, theta_s, act0, act1, lane, in, d);
  if (idx < n) {
    float *o = ders + idx * kStateDim;
    o[0] = fmaf(cpsi, s[4], -(spsi * s[5]));
    o[1] = fmaf(spsi, s[4], cpsi * s[5]);
    o[2] = negate_yaw_der ? -s[6] : s[6];
    o[3] = d[0]; o[4] = d[1]; o[5] = d[2]; o[6] = d[3];
  }
}

static NetDev to_dev(const NetDesc &n)
{
  NetDev d;
  d.n_layers = n.n_layers;
  for (int i = 0; i < 8; i++) d.layers[i] = n.layers[i];
  d.max_width = n.max_width;
  d.num_params = n.num_params;
  return d;
}

size_t valu_lds_bytes(const NetDesc &net)
{
  return ((size_t)((net.num_params + 3) & ~3) + 2 * (size_t)net.max_width * kValuLanes) * sizeof(float);
}

hipError_t launch_rollout_valu(const NetDesc &net, const RolloutArgs &a, hipStream_t stream)
{
  const size_t lds = valu_lds_bytes(net);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rollout_valu_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(rollout_valu_kernel, dim3(a.K / kValuLanes), dim3(kValuLanes), lds, stream, a,
                     to_dev(net));
  return hipGetLastError();
}

hipError_t launch_dynamics_valu(const NetDesc &net, const float *theta, const float *states,
                                const float *controls, float *ders, int n, int negate_yaw_der,
                                hipStream_t stream)
{
  const size_t lds = valu_lds_bytes(net);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dynamics_valu_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dynamics_valu_kernel, dim3((n + kValuLanes - 1) / kValuLanes), dim3(kValuLanes),
                     lds, stream, to_dev(net), theta, states, controls, ders, n, negate_yaw_der);
  return hipGetLastError();
}

}  // namespace mppi
