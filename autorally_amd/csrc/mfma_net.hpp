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
#ifdef MPPI_DIAG_NOLAST  // diagnostic build: what do the output layer's KSH matrix instructions cost? (results are garbage)
  // (every activation stays live: four fma chains over the lane's MT * 4 activations -- about what a per-lane partial of
  // a butterfly output layer would cost)
#pragma unroll
  for (int i = 0; i < MT * 4; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) o[r] = fmaf(act[i], A[aoff + ((i + r) % KSH)], o[r]);
#else
#pragma unroll
  for (int s = 0; s < KSH; s++)
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(A[aoff + s], act[s], o, 0, 0, 0);
#endif
#pragma unroll
  for (int r = 0; r < 4; r++) d[r] = o[r] + Bi[bl + r];
}

// piece 3, TREE form (rollout_multi.hip "multi4_tree"): the output layer as a reduction over the four lanes of a rollout
// instead of KSH matrix instructions on a 16-row tile of which 4 rows are outputs (8 of 28 MFMAs per step at 6-32-32-4,
// 16 of 88 at 6-64-64-4).  Lane (j, g) holds the activations of neurons 4 s + g, s = 0 .. KSH-1 (k-slot g of k-step s): it
// multiplies them into the four outputs -- a product and KSH - 1 fused multiply-adds each, s ascending, packed in pairs --
// and the four lanes' partials are summed by the halving butterfly of rollout_m44.hip: v_permlane32_swap (g < 2 keeps
// outputs {0, 1}), v_permlane16_swap (even g keeps the first of the pair).  Lane (j, g) ends with output g = the state
// component layer 0 takes from it.  NOT the reference's summation order (test oracle: fma_mode 4).
//   wt[2 s] = (W3[0][4s+g], W3[1][4s+g]), wt[2 s + 1] = (W3[2][4s+g], W3[3][4s+g]); bo = b3[g]
template <int H, int NHID>
struct MfmaTree {
  static constexpr int KSH = H / 4;
  static constexpr int nT = 4 * KSH + 1;  // floats per lane behind the MFMA image in wpack
};
template <int H, int NHID>
__device__ __forceinline__ void load_tree_weights(const float *wpack, int lane, f32x2 (&wt)[2 * MfmaTree<H, NHID>::KSH], float &bo)
{
  constexpr int KSH = MfmaTree<H, NHID>::KSH;
  const float *t = wpack + MfmaNet<H, NHID>::nPack * 64 + lane;
#pragma unroll
  for (int s = 0; s < KSH; s++) {
    wt[2 * s] = f32x2{t[(4 * s + 0) * 64], t[(4 * s + 1) * 64]};
    wt[2 * s + 1] = f32x2{t[(4 * s + 2) * 64], t[(4 * s + 3) * 64]};
  }
  bo = t[4 * KSH * 64];
}
template <int H, int NHID>
__device__ __forceinline__ float nn_last_tree(const f32x2 (&wt)[2 * MfmaTree<H, NHID>::KSH], float bo,
                                              const float (&Bi)[MfmaNet<H, NHID>::nBias],
                                              const f32x4 (&acc)[MfmaNet<H, NHID>::MT])
{
  using N = MfmaNet<H, NHID>;
  constexpr int MT = N::MT, KSH = N::KSH;
  const int boff = (NHID - 1) * MT * 4;
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
  f32x2 Q = wt[0] * f32x2{act[0], act[0]}, P = wt[1] * f32x2{act[0], act[0]};
#pragma unroll
  for (int s = 1; s < KSH; s++) {
    Q = __builtin_elementwise_fma(wt[2 * s], f32x2{act[s], act[s]}, Q);
    P = __builtin_elementwise_fma(wt[2 * s + 1], f32x2{act[s], act[s]}, P);
  }
  {
    auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(Q.x), __float_as_uint(P.x), false, false);
    auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(Q.y), __float_as_uint(P.y), false, false);
    Q = f32x2{__uint_as_float(x[0]), __uint_as_float(y[0])};
    P = f32x2{__uint_as_float(x[1]), __uint_as_float(y[1])};
  }
  const f32x2 r = Q + P;
  auto z = __builtin_amdgcn_permlane16_swap(__float_as_uint(r.x), __float_as_uint(r.y), false, false);
  return (__uint_as_float(z[0]) + __uint_as_float(z[1])) + bo;
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
