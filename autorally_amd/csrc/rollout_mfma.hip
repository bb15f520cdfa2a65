// rollout_mfma.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) for gfx950, MFMA form.
//
// Mapping (MI355X-first, not the reference's (BX,BY) thread grid):
//   * one wavefront = 16 rollouts = the N dimension of v_mfma_f32_16x16x4_f32;
//     lane l = (j = l & 15 : rollout in the wave, g = l >> 4 : k-slot / row group).
//   * every layer is D[out x 16] = W[out x in] * act[in x 16], k ascending, C = 0, bias added
//     afterwards -- bit for bit the fmaf chain of neural_net_model.cu:379-394 because the f32
//     MFMA is an in-order fmaf chain (MI355X guide, "FP32-input MFMA").
//   * the weights never leave registers: each lane holds its A-operand slice of every layer
//     (28 VGPRs for 6-32-32-4).  Row/neuron permutations chosen on the host
//     (pack_mfma_weights in mppi_abi.hip) make layer l's D registers directly the B operands
//     of layer l+1: D row 16m+4g+r carries neuron 16m+4r+g, which is k-slot g of k-step 4m+r.
//     No LDS, no cross-lane traffic, no barriers in the T loop (the reference has 8 per step).
//   * the last layer's 4 outputs are replicated in all four row groups, so each of the 4 lanes
//     of a rollout holds the full 7-float state redundantly.
//   * noise/control buffer is time-major [T][K][2]: a wave's 16 rollouts read/write one
//     contiguous 128-B line per step; the weighted reduction later streams it row by row.
#include "mppi_device.hpp"

namespace mppi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int H, int NHID>
struct MfmaNet {
  static constexpr int MT = H / 16;   // 16-row M tiles per hidden layer
  static constexpr int KSH = H / 4;   // k-steps over H inputs
  static constexpr int nA0 = MT * 2;  // layer 0: 6 inputs padded to 8 = 2 k-steps
  static constexpr int nAH = MT * KSH;
  static constexpr int nAL = KSH;     // last layer: one M tile (4 outputs x 4 row groups)
  static constexpr int nA = nA0 + (NHID - 1) * nAH + nAL;
  static constexpr int nBias = NHID * MT * 4 + 4;
  static constexpr int nPack = nA + nBias;  // floats per lane in wpack
};

// d[0..3] = NN(s3..s6, u0, u1) for the lane's rollout; every lane of the rollout gets all four.
template <int H, int NHID>
__device__ __forceinline__ void nn_forward_mfma(const float (&A)[MfmaNet<H, NHID>::nA],
                                                const float (&Bi)[MfmaNet<H, NHID>::nBias], int g,
                                                float s3, float s4, float s5, float s6, float u0,
                                                float u1, float (&d)[4])
{
  using N = MfmaNet<H, NHID>;
  constexpr int MT = N::MT, KSH = N::KSH;
  // layer-0 B operands: k-step 0 = [s3,s4,s5,s6][g], k-step 1 = [u0,u1,0,0][g]
  const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
  const float b1 = (g == 0) ? u0 : (g == 1) ? u1 : 0.0f;
  f32x4 acc[MT];
  float act[MT * 4];
#pragma unroll
  for (int m = 0; m < MT; m++) {
    f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    z = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 2 + 0], b0, z, 0, 0, 0);
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 2 + 1], b1, z, 0, 0, 0);
  }
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int r = 0; r < 4; r++) act[m * 4 + r] = tanh_fast(acc[m][r] + Bi[m * 4 + r]);

#pragma unroll
  for (int l = 1; l < NHID; l++) {
    const int aoff = N::nA0 + (l - 1) * N::nAH;
    const int boff = l * MT * 4;
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < KSH; s++)
#pragma unroll
      for (int m = 0; m < MT; m++)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[aoff + m * KSH + s], act[s], acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) act[m * 4 + r] = tanh_fast(acc[m][r] + Bi[boff + m * 4 + r]);
  }
  {
    const int aoff = N::nA0 + (NHID - 1) * N::nAH;
    const int boff = NHID * MT * 4;
    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < KSH; s++)
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(A[aoff + s], act[s], o, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) d[r] = o[r] + Bi[boff + r];  // last layer: no non-linearity
  }
}

template <int H, int NHID>
__global__ __launch_bounds__(256) void rollout_mfma_kernel(const RolloutArgs a)
{
  using N = MfmaNet<H, NHID>;
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (wave * kRolloutsPerWave >= a.K) return;  // whole wave (K is a multiple of 64)
  const int j = lane & 15, g = lane >> 4;
  const int k = wave * kRolloutsPerWave + j;

  float A[N::nA], Bi[N::nBias];
#pragma unroll
  for (int i = 0; i < N::nA; i++) A[i] = a.wpack[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < N::nBias; i++) Bi[i] = a.wpack[(N::nA + i) * 64 + lane];

  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = a.state[i];
  int crash = 0;
  float J = 0.0f;

  const int K = a.K, T = a.T;
  float2 *const noise = reinterpret_cast<float2 *>(a.noise);
  const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
  const bool noise_free_k = (k == 0);       // mppi_controller.cu:136
  const bool pure_noise_k = (k >= a.k99);   // :141, k >= .99*NUM_ROLLOUTS in double (host)

  float2 eps = noise[(size_t)k];            // t = 0
  for (int t = 0; t < T; t++) {
    const float2 e = eps;
    if (t + 1 < T) eps = noise[(size_t)(t + 1) * K + k];  // prefetch next step's line
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
    if (g == 0) noise[(size_t)t * K + k] = make_float2(u0, u1);  // stored before the clamp (Q3)
    u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
    u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);

    float spsi, cpsi;
    sincosf(s[2], &spsi, &cpsi);
    if (t > 0) {  // running mean over t = 1..T-1 of the cost of the state before the update (Q5)
      const float c = compute_cost(a.cost, a.nu, s, cpsi, spsi, u0, u1, du0, du1, crash);
      J = running_mean(J, c, t);
    }
    // computeKinematics, neural_net_model.cu:346-355
    float sd[kStateDim];
    sd[0] = fmaf(cpsi, s[4], -(spsi * s[5]));
    sd[1] = fmaf(spsi, s[4], cpsi * s[5]);
    sd[2] = a.negate_yaw_der ? -s[6] : s[6];
    float d[4];
    nn_forward_mfma<H, NHID>(A, Bi, g, s[3], s[4], s[5], s[6], u0, u1, d);
    sd[3] = d[0]; sd[4] = d[1]; sd[5] = d[2]; sd[6] = d[3];
    // incrementState, :334-344
#pragma unroll
    for (int i = 0; i < kStateDim; i++) s[i] = fmaf(sd[i], a.dt, s[i]);
    if (fabsf(s[3]) >= kRollCrash) crash = 1;  // getCrash, costs.cu:301-305
  }
  if (g == 0) a.costs[k] = J + 0.0f;  // + terminalCost (= 0), costs.cu:411-414
}

// Debug/test entry: state derivative of n independent (state, control) pairs through the
// same device functions as the rollout (used to check the golden vectors on the GPU).
template <int H, int NHID>
__global__ __launch_bounds__(64) void dynamics_mfma_kernel(const float *wpack, const float *states,
                                                           const float *controls, float *ders, int n,
                                                           int negate_yaw_der)
{
  using N = MfmaNet<H, NHID>;
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, g = lane >> 4;
  float A[N::nA], Bi[N::nBias];
#pragma unroll
  for (int i = 0; i < N::nA; i++) A[i] = wpack[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < N::nBias; i++) Bi[i] = wpack[(N::nA + i) * 64 + lane];
  const int idx = blockIdx.x * kRolloutsPerWave + j;
  const int src = idx < n ? idx : n - 1;
  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = states[src * kStateDim + i];
  const float u0 = controls[src * 2], u1 = controls[src * 2 + 1];
  float spsi, cpsi;
  sincosf(s[2], &spsi, &cpsi);
  float d[4];
  nn_forward_mfma<H, NHID>(A, Bi, g, s[3], s[4], s[5], s[6], u0, u1, d);
  if (g == 0 && idx < n) {
    float *o = ders + idx * kStateDim;
    o[0] = fmaf(cpsi, s[4], -(spsi * s[5]));
    o[1] = fmaf(spsi, s[4], cpsi * s[5]);
    o[2] = negate_yaw_der ? -s[6] : s[6];
    o[3] = d[0]; o[4] = d[1]; o[5] = d[2]; o[6] = d[3];
  }
}

// ---- host-visible launchers (declared in mppi_kernels.hpp) ----
template <int H, int NHID>
static hipError_t launch_rollout_t(const RolloutArgs &a, int block_threads, hipStream_t stream)
{
  const int waves = a.K / kRolloutsPerWave;
  const int wpb = block_threads / 64;
  const int grid = (waves + wpb - 1) / wpb;
  hipLaunchKernelGGL((rollout_mfma_kernel<H, NHID>), dim3(grid), dim3(block_threads), 0, stream, a);
  return hipGetLastError();
}

bool mfma_variant_supported(int hidden, int n_hidden)
{
  return (hidden == 32 || hidden == 64) && (n_hidden == 2 || n_hidden == 4);
}

int mfma_pack_floats_per_lane(int hidden, int n_hidden)
{
  const int MT = hidden / 16, KSH = hidden / 4;
  return MT * 2 + (n_hidden - 1) * MT * KSH + KSH + n_hidden * MT * 4 + 4;
}

hipError_t launch_rollout_mfma(int hidden, int n_hidden, const RolloutArgs &a, int block_threads,
                               hipStream_t stream)
{
  if (hidden == 32 && n_hidden == 2) return launch_rollout_t<32, 2>(a, block_threads, stream);
  if (hidden == 64 && n_hidden == 2) return launch_rollout_t<64, 2>(a, block_threads, stream);
  if (hidden == 32 && n_hidden == 4) return launch_rollout_t<32, 4>(a, block_threads, stream);
  if (hidden == 64 && n_hidden == 4) return launch_rollout_t<64, 4>(a, block_threads, stream);
  return hipErrorInvalidValue;
}

hipError_t launch_dynamics_mfma(int hidden, int n_hidden, const float *wpack, const float *states,
                                const float *controls, float *ders, int n, int negate_yaw_der,
                                hipStream_t stream)
{
  const int grid = (n + kRolloutsPerWave - 1) / kRolloutsPerWave;
#define MPPI_DYN(HH, NN)                                                                          \
  if (hidden == HH && n_hidden == NN) {                                                           \
    hipLaunchKernelGGL((dynamics_mfma_kernel<HH, NN>), dim3(grid), dim3(64), 0, stream, wpack,    \
                       states, controls, ders, n, negate_yaw_der);                                \
    return hipGetLastError();                                                                     \
  }
  MPPI_DYN(32, 2)
  MPPI_DYN(64, 2)
  MPPI_DYN(32, 4)
  MPPI_DYN(64, 4)
#undef MPPI_DYN
  return hipErrorInvalidValue;
}

}  // namespace mppi
