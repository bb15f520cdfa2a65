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
#include "noise_device.hpp"

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
      for (int r = 0; r < 4; r++) act[m * 4 + r] = tanh_bias(acc[m][r], Bi[boff + m * 4 + r]);
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
    for (int r = 0; r < 4; r++) act[m * 4 + r] = tanh_bias(acc[m][r], Bi[boff + m * 4 + r]);
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

template <int H, int NHID, bool AFFINE, bool CTRL>
__global__ __launch_bounds__(256) void rollout_mfma_kernel(const RolloutArgs a)
{
  using N = MfmaNet<H, NHID>;
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (wave * kRolloutsPerWave >= a.K) return;  // whole wave (K is a multiple of 64)
  const int j = lane & 15, g = lane >> 4;
  const int k = wave * kRolloutsPerWave + j;

  float A[N::nA], Bi[N::nBias];
  load_weights<H, NHID>(a.wpack, lane, A, Bi);

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

  // Everything a step reads from memory is requested ahead of its use (noise line, nominal
  // control and 1/t one step ahead; the two costmap texels two network layers ahead), so no
  // load latency sits on the T-step recurrence.  A step is branch-free and cut into four
  // scheduling regions: in each, the MFMAs of one network piece run next to independent
  // cost / kinematics arithmetic.
  float2 eps = noise[(size_t)k];            // t = 0
  float2 Unext = Useq[0];
  double rt_next = a.inv_t[0];
  for (int t = 0; t < T; t++) {
    // ---- region 1: controls, layer 0, sin/cos, costmap addresses and fetches ----
    const float2 e = eps;
    const float2 Ut = Unext;
    const double rt = rt_next;
    const int tn = min(t + 1, T - 1);
    eps = noise[(size_t)tn * K + k];
    Unext = Useq[tn];
    rt_next = a.inv_t[tn];
    // control perturbation, mppi_controller.cu:136-153
    const bool nf = noise_free_k | (t < a.opt_delay);
    const float n0 = e.x * a.nu[0], n1 = e.y * a.nu[1];
    const float du0 = nf ? 0.0f : n0, du1 = nf ? 0.0f : n1;
    float u0 = nf ? Ut.x : (pure_noise_k ? n0 : Ut.x + n0);
    float u1 = nf ? Ut.y : (pure_noise_k ? n1 : Ut.y + n1);
    // stored before the clamp (Q3); the four lanes of a rollout write the same value
    noise[(size_t)t * K + k] = make_float2(u0, u1);
    u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
    u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
    f32x4 acc[N::MT];
    nn_layer0<H, NHID>(A, g, s[3], s[4], s[5], s[6], u0, u1, acc);
    float spsi, cpsi;
    sincos_fast(s[2], spsi, cpsi);
    float tf, tb;
    track_fetch<AFFINE>(a.cost, s, cpsi, spsi, tf, tb);
    __builtin_amdgcn_sched_barrier(0);

    // ---- region 2: hidden layers next to kinematics and the texel-free cost terms ----
    nn_hidden<H, NHID>(A, Bi, acc);
    float sd[kStateDim];
    sd[0] = fmaf(cpsi, s[4], -(spsi * s[5]));  // computeKinematics, neural_net_model.cu:346-355
    sd[1] = fmaf(spsi, s[4], cpsi * s[5]);
    sd[2] = a.negate_yaw_der ? -s[6] : s[6];
    CostTerms ct;
    cost_terms_a<CTRL>(a.cost, a.nu, s[4], s[5], u0, u1, du0, du1, ct);
    __builtin_amdgcn_sched_barrier(0);

    // ---- region 3: output layer next to the track / crash terms and the running mean ----
    float d[4];
    nn_last<H, NHID>(A, Bi, acc, d);
    {
      // running mean over t = 1..T-1 of the cost of the state before the update (Q5); the
      // t = 0 evaluation is computed and discarded
      int crash_new = crash;
      const float c = cost_terms_b(a.cost, ct, tf, tb, crash_new);
      const float Jn = running_mean(J, c, t, rt);
      J = (t > 0) ? Jn : J;
      crash = (t > 0) ? crash_new : crash;
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- region 4: incrementState (:334-344) and getCrash (costs.cu:301-305) ----
    sd[3] = d[0]; sd[4] = d[1]; sd[5] = d[2]; sd[6] = d[3];
#pragma unroll
    for (int i = 0; i < kStateDim; i++) s[i] = fmaf(sd[i], a.dt, s[i]);
    crash |= (int)(fabsf(s[3]) >= kRollCrash);
  }
  if (g == 0) a.costs[k] = J + 0.0f;  // + terminalCost (= 0), costs.cu:411-414
}

// ---------------------------------------------------------------------------------------------
// rollout_split_kernel: the same rollout with the work of a 16-rollout group split over THREE
// wavefronts of one workgroup (they land on three SIMDs of a CU; at K = 4096 most of the chip's
// SIMDs are idle anyway):
//   wave 0 "dynamics": controls + clamp + network + Euler update of [roll, u_x, u_y, yaw_mder].
//          The learned dynamics do not depend on x, y, yaw, so this wave IS the T-step recurrence
//          and nothing else sits on it.
//   wave 1 "cost":     x, y, yaw kinematics, sin/cos, the two costmap fetches, MPPICosts::computeCost,
//          the running mean and the crash flags -- consuming the per-step record
//          (s3..s6 before the update, clamped u, du) that wave 0 leaves in an LDS ring.
//   wave 2 "noise":    the control noise of the group's 16 rollouts (MRG32k3a + Box-Muller, one lane
//          per rollout, the generator state goes HBM -> registers -> HBM once per launch), one phase
//          ahead of wave 0, through a second LDS ring: eps never touches HBM.  With explicit noise
//          (parity tests) this wave idles and wave 0 reads eps from the buffer.
// Both rings hold two phases of kPhaseSteps steps; one workgroup barrier per phase (not per step)
// hands a phase over, so the three waves run concurrently.  Arithmetic and its order are exactly
// those of rollout_mfma_kernel and noise_kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr int kPhaseSteps = 10;

template <int H, int NHID, bool AFFINE, bool CTRL>
__global__ __launch_bounds__(192) void rollout_split_kernel(const RolloutArgs a)
{
  using N = MfmaNet<H, NHID>;
  __shared__ __attribute__((aligned(16))) float ring[2][kPhaseSteps][kRolloutsPerWave][8];
  __shared__ __attribute__((aligned(16))) float2 eps_ring[2][kPhaseSteps][kRolloutsPerWave];
  const int lane = threadIdx.x & 63;
  const int role = threadIdx.x >> 6;  // wave-uniform
  const int j = lane & 15, g = lane >> 4;
  const int k = blockIdx.x * kRolloutsPerWave + j;
  const int K = a.K, T = a.T;
  const int phases = (T + kPhaseSteps - 1) / kPhaseSteps;
  // Barrier schedule (every wave executes exactly phases + 1 barriers):
  //   noise:    produce eps(p) ; barrier #(p+1)            ... then one trailing barrier
  //   dynamics: barrier #1 ; [eps(p) -> rec(p)] ; barrier #(p+2)
  //   cost:     barrier #1 ; barrier #(p+2) ; consume rec(p)

  if (role == 2) {
    // -------------------------------- noise wave --------------------------------
    const bool active = a.inline_noise && lane < kRolloutsPerWave;
    Mrg gsta{0, 0, 0, 0, 0, 0};
    if (active) {
      gsta.s10 = a.rng_in[k]; gsta.s11 = a.rng_in[K + k]; gsta.s12 = a.rng_in[2 * K + k];
      gsta.s20 = a.rng_in[3 * K + k]; gsta.s21 = a.rng_in[4 * K + k]; gsta.s22 = a.rng_in[5 * K + k];
    }
    for (int p = 0; p < phases; p++) {
      const int nq = min(kPhaseSteps, T - p * kPhaseSteps);
      if (active)
        for (int q = 0; q < nq; q++) eps_ring[p & 1][q][lane] = noise_pair(gsta);
      __syncthreads();
    }
    if (active) {
      a.rng_out[k] = gsta.s10; a.rng_out[K + k] = gsta.s11; a.rng_out[2 * K + k] = gsta.s12;
      a.rng_out[3 * K + k] = gsta.s20; a.rng_out[4 * K + k] = gsta.s21; a.rng_out[5 * K + k] = gsta.s22;
    }
    __syncthreads();
  } else if (role == 0) {
    // ------------------------------ dynamics wave ------------------------------
    float A[N::nA], Bi[N::nBias];
    load_weights<H, NHID>(a.wpack, lane, A, Bi);
    float s3 = a.state[3], s4 = a.state[4], s5 = a.state[5], s6 = a.state[6];
    float2 *const noise = reinterpret_cast<float2 *>(a.noise);
    const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
    const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
    const bool pure_noise_k = (k >= a.k99);  // :141
    const bool inl = a.inline_noise != 0;
    float2 eps = *(inl ? Useq : &noise[(size_t)k]);
    float2 Unext = Useq[0];
    __syncthreads();  // barrier #1: eps(0) is in the ring
    float2 el_next = eps_ring[0][0][j];  // ring value of the step about to run (read one step ahead)
    for (int p = 0; p < phases; p++) {
      const int t0 = p * kPhaseSteps, nq = min(kPhaseSteps, T - t0);
      for (int q = 0; q < nq; q++) {
        const int t = t0 + q;
        const float2 eg = eps;  // explicit-noise path: requested one step ahead from the buffer
        const float2 el = el_next;
        const float2 Ut = Unext;
        const int tn = min(t + 1, T - 1);
        // always one load (so the compiler can count outstanding loads and never drains the queue): the
        // buffer with explicit noise, a cache-resident dummy (no HBM read of eps) with in-kernel noise
        eps = *(inl ? Useq : &noise[(size_t)tn * K + k]);
        Unext = Useq[tn];
        if (q + 1 < nq) el_next = eps_ring[p & 1][q + 1][j];
        const float2 e = inl ? el : eg;
        // control perturbation, mppi_controller.cu:136-153
        const bool nf = noise_free_k | (t < a.opt_delay);
        const float n0 = e.x * a.nu[0], n1 = e.y * a.nu[1];
        const float du0 = nf ? 0.0f : n0, du1 = nf ? 0.0f : n1;
        float u0 = nf ? Ut.x : (pure_noise_k ? n0 : Ut.x + n0);
        float u1 = nf ? Ut.y : (pure_noise_k ? n1 : Ut.y + n1);
        noise[(size_t)t * K + k] = make_float2(u0, u1);  // before the clamp (Q3)
        // pin the two prefetches above: issued here, first used at the top of the NEXT step, so
        // their latency never sits on the recurrence (the scheduler otherwise sinks them)
        __builtin_amdgcn_sched_barrier(0);
        u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
        u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
        // record for the cost wave: state BEFORE this step's update, clamped u, du
        float2 rec;
        rec.x = (g == 0) ? s3 : (g == 1) ? s5 : (g == 2) ? u0 : du0;
        rec.y = (g == 0) ? s4 : (g == 1) ? s6 : (g == 2) ? u1 : du1;
        *reinterpret_cast<float2 *>(&ring[p & 1][q][j][2 * g]) = rec;
        float d[4];
        nn_forward_mfma<H, NHID>(A, Bi, g, s3, s4, s5, s6, u0, u1, d);
        s3 = fmaf(d[0], a.dt, s3);  // incrementState, neural_net_model.cu:334-344
        s4 = fmaf(d[1], a.dt, s4);
        s5 = fmaf(d[2], a.dt, s5);
        s6 = fmaf(d[3], a.dt, s6);
      }
      __syncthreads();  // barrier #(p+2): rec(p) is complete, eps(p+1) is in the ring
      el_next = eps_ring[(p + 1) & 1][0][j];
    }
  } else {
    // -------------------------------- cost wave --------------------------------
    float x = a.state[0], y = a.state[1], yaw = a.state[2];
    int crash = 0;
    float J = 0.0f;
    double rt_next = a.inv_t[0];
    __syncthreads();  // barrier #1
    for (int p = 0; p < phases; p++) {
      __syncthreads();  // barrier #(p+2): rec(p) is complete
      const int t0 = p * kPhaseSteps, nq = min(kPhaseSteps, T - t0);
      for (int q = 0; q < nq; q++) {
        const int t = t0 + q;
        const double rt = rt_next;
        rt_next = a.inv_t[min(t + 1, T - 1)];
        const float4 r0 = *reinterpret_cast<const float4 *>(&ring[p & 1][q][j][0]);  // s3 s4 s5 s6
        const float4 r1 = *reinterpret_cast<const float4 *>(&ring[p & 1][q][j][4]);  // u0 u1 du0 du1
        // getCrash of the previous step's update (costs.cu:301-305): r0.x is s3 after update t-1
        crash |= (int)((t > 0) & (fabsf(r0.x) >= kRollCrash));
        float spsi, cpsi;
        sincos_fast(yaw, spsi, cpsi);
        const float st[3] = {x, y, yaw};
        float tf, tb;
        track_fetch<AFFINE>(a.cost, st, cpsi, spsi, tf, tb);
        CostTerms ct;
        cost_terms_a<CTRL>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
        // computeKinematics + incrementState for x, y, yaw (neural_net_model.cu:346-355, 334-344)
        const float sd0 = fmaf(cpsi, r0.y, -(spsi * r0.z));
        const float sd1 = fmaf(spsi, r0.y, cpsi * r0.z);
        const float sd2 = a.negate_yaw_der ? -r0.w : r0.w;
        x = fmaf(sd0, a.dt, x);
        y = fmaf(sd1, a.dt, y);
        yaw = fmaf(sd2, a.dt, yaw);
        // running mean over t = 1..T-1 (Q5); the t = 0 evaluation is discarded
        int crash_new = crash;
        const float c = cost_terms_b(a.cost, ct, tf, tb, crash_new);
        const float Jn = running_mean(J, c, t, rt);
        J = (t > 0) ? Jn : J;
        crash = (t > 0) ? crash_new : crash;
      }
    }
    a.costs[k] = J + 0.0f;  // + terminalCost (= 0); the 4 lanes of a rollout write the same value
  }
}

// ---------------------------------------------------------------------------------------------
// rollout_quad_kernel: FOUR wavefronts per 16 rollouts -- the network itself is split over two
// SIMDs.  Used while 4*K/16 <= number of SIMDs (K <= 4096 on MI355X), where every wave owns a SIMD.
//
// Why: the f32 MFMA occupies the f32 vector datapath, so one wave cannot go below
// (MFMA cycles + VALU cycles) per step.  Halving the recurrence needs a second datapath:
//   wave 0 / wave 1 "dynamics": each owns half of the 16-row M tiles of every hidden layer (half of
//          the MFMAs and half of the tanh of layers 0..NHID-1).  Before each following layer the
//          two waves swap their activations through LDS -- lane l of one wave needs exactly
//          registers r of lane l of the other (same rollout j, same k-slot g), so the swap is one
//          16-byte store + one 16-byte load per lane -- and one workgroup barrier.  The output
//          layer (a single M tile, a serial chain) is computed by both, so both hold the new state.
//          Every dot product keeps its k-ascending order: bit-identical results.
//   wave 2 "cost", wave 3 "noise": as in rollout_split_kernel, but in lock step with the NHID
//          barriers per step; their work is cut at the barriers so that neither delays one.
// ---------------------------------------------------------------------------------------------
template <int H, int NHID, bool AFFINE, bool CTRL>
__global__ __launch_bounds__(256) void rollout_quad_kernel(const RolloutArgs a)
{
  using N = MfmaNet<H, NHID>;
  constexpr int MT = N::MT, M2 = MT / 2, KSH = N::KSH, NX = M2 * 4;
  static_assert(MT % 2 == 0 && NHID >= 2, "needs two M tiles and two exchanges per step");
  __shared__ __attribute__((aligned(16))) float xb[NHID][2][64][NX];
  __shared__ __attribute__((aligned(16))) float rec[2][kRolloutsPerWave][8];
  __shared__ __attribute__((aligned(16))) float2 eps_ring[3][kRolloutsPerWave];
  const int lane = threadIdx.x & 63;
  const int role = threadIdx.x >> 6;  // wave-uniform
  const int j = lane & 15, g = lane >> 4;
  const int k = blockIdx.x * kRolloutsPerWave + j;
  const int K = a.K, T = a.T;
  const bool inl = a.inline_noise != 0;
  // Every wave executes exactly 1 + NHID*T barriers: the prologue barrier, then barriers
  // B_0(t) .. B_{NHID-1}(t) of step t (one per activation swap).

  if (role == 3) {
    // -------------------------------- noise wave --------------------------------
    // eps(t+2) is drawn during step t (generator steps after B_0, Box-Muller after B_1) into a
    // three-slot ring; the dynamics waves read eps(t+1) after B_0(t).
    const bool active = inl && lane < kRolloutsPerWave;
    Mrg gsta{0, 0, 0, 0, 0, 0};
    if (active) {
      gsta.s10 = a.rng_in[k]; gsta.s11 = a.rng_in[K + k]; gsta.s12 = a.rng_in[2 * K + k];
      gsta.s20 = a.rng_in[3 * K + k]; gsta.s21 = a.rng_in[4 * K + k]; gsta.s22 = a.rng_in[5 * K + k];
      eps_ring[0][lane] = noise_pair(gsta);
      if (T > 1) eps_ring[1][lane] = noise_pair(gsta);
    }
    __syncthreads();  // prologue
    for (int t = 0; t < T; t++) {
      const bool draw = active && (t + 2 < T);
      __syncthreads();  // B_0(t)
      float u1 = 0.0f, u2 = 0.0f;
      if (draw) {
        u1 = (float)mrg_next_z(gsta) * 0x1p-32f;
        u2 = (float)mrg_next_z(gsta) * 0x1p-32f;
      }
      __syncthreads();  // B_1(t)
      if (draw) {
        const float r = sqrtf(-2.0f * spec_logf(u1));
        float sn, cs;
        spec_sincos2pi(u2, sn, cs);
        eps_ring[(t + 2) % 3][lane] = make_float2(r * sn, r * cs);
      }
#pragma unroll
      for (int e = 2; e < NHID; e++) __syncthreads();
    }
    if (active) {
      a.rng_out[k] = gsta.s10; a.rng_out[K + k] = gsta.s11; a.rng_out[2 * K + k] = gsta.s12;
      a.rng_out[3 * K + k] = gsta.s20; a.rng_out[4 * K + k] = gsta.s21; a.rng_out[5 * K + k] = gsta.s22;
    }
  } else if (role == 2) {
    // -------------------------------- cost wave --------------------------------
    float x = a.state[0], y = a.state[1], yaw = a.state[2];
    int crash = 0;
    float J = 0.0f;
    double rt_next = a.inv_t[0];
    __syncthreads();  // prologue
    for (int t = 0; t < T; t++) {
      const double rt = rt_next;
      rt_next = a.inv_t[min(t + 1, T - 1)];
      __syncthreads();  // B_0(t): rec(t) is in LDS
      const float4 r0 = *reinterpret_cast<const float4 *>(&rec[t & 1][j][0]);  // s3 s4 s5 s6
      const float4 r1 = *reinterpret_cast<const float4 *>(&rec[t & 1][j][4]);  // u0 u1 du0 du1
      crash |= (int)((t > 0) & (fabsf(r0.x) >= kRollCrash));  // getCrash of update t-1
      float spsi, cpsi;
      sincos_fast(yaw, spsi, cpsi);
      const float st[3] = {x, y, yaw};
      float tf, tb;
      track_fetch<AFFINE>(a.cost, st, cpsi, spsi, tf, tb);
      const float sd0 = fmaf(cpsi, r0.y, -(spsi * r0.z));
      const float sd1 = fmaf(spsi, r0.y, cpsi * r0.z);
      const float sd2 = a.negate_yaw_der ? -r0.w : r0.w;
      x = fmaf(sd0, a.dt, x);
      y = fmaf(sd1, a.dt, y);
      yaw = fmaf(sd2, a.dt, yaw);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // B_1(t)
      CostTerms ct;
      cost_terms_a<CTRL>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
      int crash_new = crash;
      const float c = cost_terms_b(a.cost, ct, tf, tb, crash_new);
      const float Jn = running_mean(J, c, t, rt);
      J = (t > 0) ? Jn : J;
      crash = (t > 0) ? crash_new : crash;
#pragma unroll
      for (int e = 2; e < NHID; e++) __syncthreads();
    }
    a.costs[k] = J + 0.0f;
  } else {
    // --------------------------- dynamics waves (w = 0, 1) ---------------------------
    const int w = role;
    float A0[M2 * 2], AH[(NHID - 1) * M2 * KSH], AL[KSH], Bh[NHID * NX], BL[4];
#pragma unroll
    for (int i = 0; i < M2; i++) {
      const int m = w * M2 + i;
#pragma unroll
      for (int s = 0; s < 2; s++) A0[i * 2 + s] = a.wpack[(m * 2 + s) * 64 + lane];
#pragma unroll
      for (int l = 1; l < NHID; l++)
#pragma unroll
        for (int s = 0; s < KSH; s++)
          AH[((l - 1) * M2 + i) * KSH + s] = a.wpack[(N::nA0 + (l - 1) * N::nAH + m * KSH + s) * 64 + lane];
#pragma unroll
      for (int l = 0; l < NHID; l++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          Bh[l * NX + i * 4 + r] = a.wpack[(N::nA + l * MT * 4 + m * 4 + r) * 64 + lane] * kTanhScale;
    }
#pragma unroll
    for (int s = 0; s < KSH; s++) AL[s] = a.wpack[(N::nA0 + (NHID - 1) * N::nAH + s) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; r++) BL[r] = a.wpack[(N::nA + NHID * MT * 4 + r) * 64 + lane];

    float s3 = a.state[3], s4 = a.state[4], s5 = a.state[5], s6 = a.state[6];
    float2 *const noise = reinterpret_cast<float2 *>(a.noise);
    const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
    const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
    const bool pure_noise_k = (k >= a.k99);  // :141
    float2 eps = *(inl ? Useq : &noise[(size_t)k]);
    float2 Unext = Useq[0];
    __syncthreads();  // prologue: eps(0), eps(1) are in the ring
    float2 el_next = eps_ring[0][j];
    for (int t = 0; t < T; t++) {
      const float2 eg = eps;
      const float2 el = el_next;
      const float2 Ut = Unext;
      const int tn = min(t + 1, T - 1);
      eps = *(inl ? Useq : &noise[(size_t)tn * K + k]);  // always one load; dummy with in-kernel noise
      Unext = Useq[tn];
      const float2 e = inl ? el : eg;
      // control perturbation, mppi_controller.cu:136-153 (computed by both dynamics waves)
      const bool nf = noise_free_k | (t < a.opt_delay);
      const float n0 = e.x * a.nu[0], n1 = e.y * a.nu[1];
      const float du0 = nf ? 0.0f : n0, du1 = nf ? 0.0f : n1;
      float u0 = nf ? Ut.x : (pure_noise_k ? n0 : Ut.x + n0);
      float u1 = nf ? Ut.y : (pure_noise_k ? n1 : Ut.y + n1);
      if (w == 0) noise[(size_t)t * K + k] = make_float2(u0, u1);  // before the clamp (Q3)
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetches above (first used next step)
      u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
      u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
      if (w == 0) {  // record for the cost wave: state BEFORE this step's update, clamped u, du
        float2 rc;
        rc.x = (g == 0) ? s3 : (g == 1) ? s5 : (g == 2) ? u0 : du0;
        rc.y = (g == 0) ? s4 : (g == 1) ? s6 : (g == 2) ? u1 : du1;
        *reinterpret_cast<float2 *>(&rec[t & 1][j][2 * g]) = rc;
      }
      // layer 0, own M tiles
      const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
      const float b1 = (g == 0) ? u0 : (g == 1) ? u1 : 0.0f;
      f32x4 acc[M2];
#pragma unroll
      for (int i = 0; i < M2; i++) {
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        z = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[i * 2 + 0], b0, z, 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[i * 2 + 1], b1, z, 0, 0, 0);
      }
      float d[4];
#pragma unroll
      for (int e2 = 0; e2 < NHID; e2++) {
        // own half of the activations of hidden layer e2, swapped with the partner wave
        float own[NX], oth[NX];
#pragma unroll
        for (int i = 0; i < M2; i++)
#pragma unroll
          for (int r = 0; r < 4; r++) own[i * 4 + r] = tanh_bias(acc[i][r], Bh[e2 * NX + i * 4 + r]);
#pragma unroll
        for (int q = 0; q < NX / 4; q++)
          *reinterpret_cast<float4 *>(&xb[e2][w][lane][4 * q]) =
              make_float4(own[4 * q], own[4 * q + 1], own[4 * q + 2], own[4 * q + 3]);
        __syncthreads();  // B_e2(t)
#pragma unroll
        for (int q = 0; q < NX / 4; q++) {
          const float4 v = *reinterpret_cast<const float4 *>(&xb[e2][1 - w][lane][4 * q]);
          oth[4 * q] = v.x; oth[4 * q + 1] = v.y; oth[4 * q + 2] = v.z; oth[4 * q + 3] = v.w;
        }
        if (e2 == 0) el_next = eps_ring[(t + 1) % 3][j];  // written during step t-1, visible after B_0(t)
        // activation of k-step s = 4m + r: from the wave that owns tile m
        float act[MT * 4];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const bool mine = (m / M2) == 0;  // tile owned by wave 0
            const int li = (m % M2) * 4 + r;
            act[m * 4 + r] = (w == 0) ? (mine ? own[li] : oth[li]) : (mine ? oth[li] : own[li]);
          }
        if (e2 < NHID - 1) {
#pragma unroll
          for (int i = 0; i < M2; i++) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int s = 0; s < KSH; s++)
#pragma unroll
            for (int i = 0; i < M2; i++)
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(AH[(e2 * M2 + i) * KSH + s], act[s], acc[i], 0, 0, 0);
        } else {
          f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int s = 0; s < KSH; s++) o = __builtin_amdgcn_mfma_f32_16x16x4f32(AL[s], act[s], o, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; r++) d[r] = o[r] + BL[r];
        }
      }
      s3 = fmaf(d[0], a.dt, s3);  // incrementState, neural_net_model.cu:334-344
      s4 = fmaf(d[1], a.dt, s4);
      s5 = fmaf(d[2], a.dt, s5);
      s6 = fmaf(d[3], a.dt, s6);
    }
  }
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
  load_weights<H, NHID>(wpack, lane, A, Bi);
  const int idx = blockIdx.x * kRolloutsPerWave + j;
  const int src = idx < n ? idx : n - 1;
  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = states[src * kStateDim + i];
  const float u0 = controls[src * 2], u1 = controls[src * 2 + 1];
  float spsi, cpsi;
  sincos_fast(s[2], spsi, cpsi);
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
  const bool affine = a.cost.affine != 0, ctrl = a.cost.need_control_cost != 0;
  if (block_threads == 512) {  // quad form: two dynamics waves + cost wave + noise wave per 16 rollouts
    const dim3 grid(a.K / kRolloutsPerWave), block(256);
    if (affine && !ctrl) hipLaunchKernelGGL((rollout_quad_kernel<H, NHID, true, false>), grid, block, 0, stream, a);
    else if (affine && ctrl) hipLaunchKernelGGL((rollout_quad_kernel<H, NHID, true, true>), grid, block, 0, stream, a);
    else if (!affine && !ctrl) hipLaunchKernelGGL((rollout_quad_kernel<H, NHID, false, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((rollout_quad_kernel<H, NHID, false, true>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  if (block_threads == 128) {  // split form: one dynamics wave + one cost wave per 16 rollouts
    const dim3 grid(a.K / kRolloutsPerWave), block(192);
    if (affine && !ctrl) hipLaunchKernelGGL((rollout_split_kernel<H, NHID, true, false>), grid, block, 0, stream, a);
    else if (affine && ctrl) hipLaunchKernelGGL((rollout_split_kernel<H, NHID, true, true>), grid, block, 0, stream, a);
    else if (!affine && !ctrl) hipLaunchKernelGGL((rollout_split_kernel<H, NHID, false, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((rollout_split_kernel<H, NHID, false, true>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  const int waves = a.K / kRolloutsPerWave;
  const int wpb = block_threads / 64;
  const dim3 grid((waves + wpb - 1) / wpb), block(block_threads);
  if (affine && !ctrl) hipLaunchKernelGGL((rollout_mfma_kernel<H, NHID, true, false>), grid, block, 0, stream, a);
  else if (affine && ctrl) hipLaunchKernelGGL((rollout_mfma_kernel<H, NHID, true, true>), grid, block, 0, stream, a);
  else if (!affine && !ctrl) hipLaunchKernelGGL((rollout_mfma_kernel<H, NHID, false, false>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((rollout_mfma_kernel<H, NHID, false, true>), grid, block, 0, stream, a);
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
