// mfma_net.hpp -- the network of NeuralNetModel::computeDynamics (neural_net_model.cu:357-410) on
// v_mfma_f32_16x16x4_f32, shared by the MFMA rollout kernels (rollout_mfma.hip, rollout_multi.hip).
//
//   * one wavefront = 16 rollouts = the N dimension of the instruction;
//     lane l = (j = l & 15 : rollout in the wave, g = l >> 4 : k-slot / row group).
//   * every layer is D[out x 16] = W[out x in] * act[in x 16], k ascending, C = 0, bias added
//     afterwards -- bit for bit the fmaf chain of neural_net_model.cu:379-394 because the f32
//     MFMA is an in-order fmaf chain (MI355X guide, "FP32-input MFMA").
//   * the weights never leave registers: each lane holds its A-operand slice of every layer
//     (28 VGPRs for 6-32-32-4).  Row/neuron permutations chosen on the host
//     (pack_mfma_weights in mppi_abi.hip) make layer l's D registers directly the B operands
//     of layer l+1: D row 16m+4g+r carries neuron 16m+4r+g, which is k-slot g of k-step 4m+r.
//     No LDS, no cross-lane traffic, no barriers for the network (the reference has 8 per step).
//   * the last layer's 4 outputs are replicated in all four row groups, so each of the 4 lanes
//     of a rollout holds the four outputs.
#pragma once

#include "mppi_device.hpp"

namespace mppi {

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

// Loads a lane's A-operand and bias slices; hidden-layer biases are pre-scaled for tanh_bias.
template <int H, int NHID>
__device__ __forceinline__ void load_weights(const float *wpack, int lane, float (&A)[MfmaNet<H, NHID>::nA],
                                             float (&Bi)[MfmaNet<H, NHID>::nBias])
{
  using N = MfmaNet<H, NHID>;
#pragma unroll
  for (int i = 0; i < N::nA; i++) A[i] = wpack[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < N::nBias; i++) {
    const float b = wpack[(N::nA + i) * 64 + lane];
    Bi[i] = (i < NHID * N::MT * 4) ? b * kTanhScale : b;
  }
}

// The network is evaluated in three pieces so that the rollout step can place independent
// cost / kinematics arithmetic next to each piece (they execute in the shadow of the MFMAs).
//
// piece 1: layer 0.  B operands: k-step 0 = [s3,s4,s5,s6][g], k-step 1 = [u0,u1,0,0][g].
template <int H, int NHID>
__device__ __forceinline__ void nn_layer0(const float (&A)[MfmaNet<H, NHID>::nA], int g, float s3,
                                          float s4, float s5, float s6, float u0, float u1,
                                          f32x4 (&acc)[MfmaNet<H, NHID>::MT])
{
  constexpr int MT = MfmaNet<H, NHID>::MT;
  const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
  const float b1 = (g == 0) ? u0 : (g == 1) ? u1 : 0.0f;
#pragma unroll
  for (int m = 0; m < MT; m++) {
    f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    z = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 2 + 0], b0, z, 0, 0, 0);
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 2 + 1], b1, z, 0, 0, 0);
  }
}

// piece 2: tanh(layer-0 output) and the hidden->hidden layers; leaves the pre-activation of the
// last hidden layer in acc.
template <int H, int NHID>
__device__ __forceinline__ void nn_hidden(const float (&A)[MfmaNet<H, NHID>::nA],
                                          const float (&Bi)[MfmaNet<H, NHID>::nBias],
                                          f32x4 (&acc)[MfmaNet<H, NHID>::MT])
{
  using N = MfmaNet<H, NHID>;
  constexpr int MT = N::MT, KSH = N::KSH;
#pragma unroll
  for (int l = 1; l < NHID; l++) {
    const int aoff = N::nA0 + (l - 1) * N::nAH;
    const int boff = (l - 1) * MT * 4;
    float act[MT * 4];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2 v = tanh_bias2(f32x2{acc[m][r], acc[m][r + 1]},
                                   f32x2{Bi[boff + m * 4 + r], Bi[boff + m * 4 + r + 1]});
        act[m * 4 + r] = v.x;
        act[m * 4 + r + 1] = v.y;
      }
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < KSH; s++)
#pragma unroll
      for (int m = 0; m < MT; m++)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[aoff + m * KSH + s], act[s], acc[m], 0, 0, 0);
  }
}

// piece 3: tanh(last hidden pre-activation), output layer (no non-linearity), bias.
template <int H, int NHID>
__device__ __forceinline__ void nn_last(const float (&A)[MfmaNet<H, NHID>::nA],
                                        const float (&Bi)[MfmaNet<H, NHID>::nBias],
                                        const f32x4 (&acc)[MfmaNet<H, NHID>::MT], float (&d)[4])
{
  using N = MfmaNet<H, NHID>;
  constexpr int MT = N::MT, KSH = N::KSH;
  const int aoff = N::nA0 + (NHID - 1) * N::nAH;
  const int boff = (NHID - 1) * MT * 4, bl = NHID * MT * 4;
  float act[MT * 4];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const f32x2 v = tanh_bias2(f32x2{acc[m][r], acc[m][r + 1]},
                                 f32x2{Bi[boff + m * 4 + r], Bi[boff + m * 4 + r + 1]});
      act[m * 4 + r] = v.x;
      act[m * 4 + r + 1] = v.y;
    }
  f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int s = 0; s < KSH; s++)
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(A[aoff + s], act[s], o, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) d[r] = o[r] + Bi[bl + r];
}

// d[0..3] = NN(s3..s6, u0, u1) for the lane's rollout; every lane of the rollout gets all four.
template <int H, int NHID>
__device__ __forceinline__ void nn_forward_mfma(const float (&A)[MfmaNet<H, NHID>::nA],
                                                const float (&Bi)[MfmaNet<H, NHID>::nBias], int g,
                                                float s3, float s4, float s5, float s6, float u0,
                                                float u1, float (&d)[4])
{
  f32x4 acc[MfmaNet<H, NHID>::MT];
  nn_layer0<H, NHID>(A, g, s3, s4, s5, s6, u0, u1, acc);
  nn_hidden<H, NHID>(A, Bi, acc);
  nn_last<H, NHID>(A, Bi, acc, d);
}

// layer 0 with the two B operands given directly: b0 = [s3,s4,s5,s6][g], b1 = [u0,u1,0,0][g]
template <int H, int NHID>
__device__ __forceinline__ void nn_layer0_ops(const float (&A)[MfmaNet<H, NHID>::nA], float b0, float b1,
                                              f32x4 (&acc)[MfmaNet<H, NHID>::MT])
{
  constexpr int MT = MfmaNet<H, NHID>::MT;
#pragma unroll
  for (int m = 0; m < MT; m++) {
    f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    z = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 2 + 0], b0, z, 0, 0, 0);
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 2 + 1], b1, z, 0, 0, 0);
  }
}

}  // namespace mppi
