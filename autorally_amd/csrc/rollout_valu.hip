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

// ---------------------------------------------------------------------------------------------
// rollout_valu_reg_kernel<H, NHID>: the vector-ALU arm of the MFMA-vs-VALU A/B (SURVEY cfg 4) for
// the standard 6 -> H x NHID -> 4 shapes.  One lane per rollout (64 rollouts per wavefront),
// activations in registers, the packed weight blob staged once into LDS and read back as 16-byte
// broadcasts (one ds_read_b128 feeds four v_fmac): a straightforward CDNA vector-ALU
// implementation of the reference's inner loop (neural_net_model.cu:379-394).
// Same k-ascending fmaf chain and the same tanh_bias as the MFMA kernels: bit-identical results.
// `theta_s` is the packed blob with the hidden-layer biases pre-multiplied by kTanhScale (host).
// ---------------------------------------------------------------------------------------------
// One dense layer: inputs in registers (the k loop is unrolled), one weight row per iteration of
// the (rolled) output loop read from LDS as broadcasts, outputs parked in a lane-major LDS tile
// (bank = lane) and pulled back into registers for the next layer.
template <int NIN, int NOUT, bool TANH>
__device__ __forceinline__ void dense_reg(const float *W, const float *b, const float (&in)[NIN],
                                          float *tile, int lane)
{
#pragma unroll 4  // four independent fmaf chains in flight (each chain is k-ascending, hence serial)
  for (int jn = 0; jn < NOUT; jn++) {
    const float *row = W + jn * NIN;
    float acc = 0.0f;
    if (NIN % 4 == 0) {
      const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
      for (int q = 0; q < NIN / 4; q++) {
        const float4 w = r4[q];
        acc = fmaf(w.x, in[4 * q + 0], acc);
        acc = fmaf(w.y, in[4 * q + 1], acc);
        acc = fmaf(w.z, in[4 * q + 2], acc);
        acc = fmaf(w.w, in[4 * q + 3], acc);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < NIN; kk++) acc = fmaf(row[kk], in[kk], acc);
    }
    tile[jn * kValuLanes + lane] = TANH ? tanh_bias(acc, b[jn]) : acc + b[jn];
  }
}

template <int H, int NHID>
__device__ __forceinline__ void nn_forward_reg(const float *theta_s, float *tile, int lane,
                                               const float (&in)[kNetIn], float (&d)[kNetOut])
{
  float a[H];
  dense_reg<kNetIn, H, true>(theta_s, theta_s + kNetIn * H, in, tile, lane);
  int off = kNetIn * H + H;
#pragma unroll
  for (int l = 1; l < NHID; l++) {
#pragma unroll
    for (int i = 0; i < H; i++) a[i] = tile[i * kValuLanes + lane];
    dense_reg<H, H, true>(theta_s + off, theta_s + off + H * H, a, tile, lane);
    off += H * H + H;
  }
#pragma unroll
  for (int i = 0; i < H; i++) a[i] = tile[i * kValuLanes + lane];
  dense_reg<H, kNetOut, false>(theta_s + off, theta_s + off + H * kNetOut, a, tile, lane);
#pragma unroll
  for (int i = 0; i < kNetOut; i++) d[i] = tile[i * kValuLanes + lane];
}

template <int H, int NHID>
__global__ __launch_bounds__(kValuLanes) void rollout_valu_reg_kernel(const RolloutArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float theta_lds[];
  const int lane = threadIdx.x;
  constexpr int kParams = (kNetIn + 1) * H + (NHID - 1) * (H + 1) * H + (H + 1) * kNetOut;
  for (int i = lane; i < kParams; i += kValuLanes) theta_lds[i] = a.wpack[i];
  __syncthreads();
  const int k = blockIdx.x * kValuLanes + lane;
  if (k >= a.K) return;
  const float *theta_s = theta_lds;
  float *tile = theta_lds + ((kParams + 3) & ~3);  // [H][64] activation tile
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
  float2 eps = noise[(size_t)k];
  for (int t = 0; t < T; t++) {
    const float2 e = eps;
    eps = noise[(size_t)min(t + 1, T - 1) * K + k];
    const float2 Ut = Useq[t];
    const bool nf = noise_free_k | (t < a.opt_delay);
    const float n0 = e.x * a.nu[0], n1 = e.y * a.nu[1];
    const float du0 = nf ? 0.0f : n0, du1 = nf ? 0.0f : n1;
    float u0 = nf ? Ut.x : (pure_noise_k ? n0 : Ut.x + n0);
    float u1 = nf ? Ut.y : (pure_noise_k ? n1 : Ut.y + n1);
    noise[(size_t)t * K + k] = make_float2(u0, u1);
    u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
    u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
    float spsi, cpsi;
    sincos_fast(s[2], spsi, cpsi);
    float tf, tb;
    if (a.cost.affine) track_fetch<true>(a.cost, s, cpsi, spsi, tf, tb);
    else track_fetch<false>(a.cost, s, cpsi, spsi, tf, tb);
    float sd[kStateDim];
    sd[0] = fmaf(cpsi, s[4], -(spsi * s[5]));
    sd[1] = fmaf(spsi, s[4], cpsi * s[5]);
    sd[2] = a.negate_yaw_der ? -s[6] : s[6];
    const float in[kNetIn] = {s[3], s[4], s[5], s[6], u0, u1};
    float d[kNetOut];
    nn_forward_reg<H, NHID>(theta_s, tile, lane, in, d);
    sd[3] = d[0]; sd[4] = d[1]; sd[5] = d[2]; sd[6] = d[3];
    {
      int crash_new = crash;
      const float c = a.cost.need_control_cost
                          ? cost_finish<true>(a.cost, a.nu, s[4], s[5], tf, tb, u0, u1, du0, du1, crash_new)
                          : cost_finish<false>(a.cost, a.nu, s[4], s[5], tf, tb, u0, u1, du0, du1, crash_new);
      const float Jn = running_mean(J, c, t, a.inv_t[t]);
      J = (t > 0) ? Jn : J;
      crash = (t > 0) ? crash_new : crash;
    }
#pragma unroll
    for (int i = 0; i < kStateDim; i++) s[i] = fmaf(sd[i], a.dt, s[i]);
    crash |= (int)(fabsf(s[3]) >= kRollCrash);
  }
  a.costs[k] = J + 0.0f;
}

bool valu_reg_supported(int hidden, int n_hidden)
{
  return (hidden == 32 || hidden == 64) && (n_hidden == 2 || n_hidden == 4);
}

hipError_t launch_rollout_valu_reg(int hidden, int n_hidden, const RolloutArgs &a, hipStream_t stream)
{
  const dim3 grid(a.K / kValuLanes), block(kValuLanes);
  const int n_params = (kNetIn + 1) * hidden + (n_hidden - 1) * (hidden + 1) * hidden + (hidden + 1) * kNetOut;
  const size_t lds = ((size_t)((n_params + 3) & ~3) + (size_t)hidden * kValuLanes) * sizeof(float);
#define MPPI_VR(HH, NN)                                                                             \
  if (hidden == HH && n_hidden == NN) {                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rollout_valu_reg_kernel<HH, NN>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
    if (e != hipSuccess) return e;                                                                  \
    MPPI_LAUNCH_ROLLOUT((rollout_valu_reg_kernel<HH, NN>), grid, block, lds, stream, a);             \
    return hipGetLastError();                                                                       \
  }
  MPPI_VR(32, 2)
  MPPI_VR(64, 2)
  MPPI_VR(32, 4)
  MPPI_VR(64, 4)
#undef MPPI_VR
  return hipErrorInvalidValue;
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
  MPPI_LAUNCH_ROLLOUT(rollout_valu_kernel, dim3(a.K / kValuLanes), dim3(kValuLanes), lds, stream, a,
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
