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
#include "mfma_net.hpp"
#include "noise_device.hpp"
#include "mppi_kernels.hpp"

namespace mppi {


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
// rollout_quad_kernel: FOUR wavefronts per 16 rollouts -- the network itself runs on two SIMDs.
// Used while 4*K/16 <= number of SIMDs (K <= 4096 on MI355X), where every wave owns a SIMD.
//
// Why: the f32 MFMA occupies the f32 vector datapath, so one wave cannot go below
// (MFMA cycles + VALU cycles) per step; halving the recurrence needs a second datapath.
//   wave 0 / wave 1 "dynamics": each owns half of the 16-row M tiles of every hidden layer after
//          the first (half of their MFMAs and tanh) and swaps its activations with the partner
//          through LDS before the next layer: lane l of one wave needs exactly the registers of
//          lane l of the other (same rollout j, same k-slot g), so a swap is one 16-byte store and
//          one 16-byte load per lane.  The output layer (a single M tile, a serial chain) is
//          computed by both, so both hold the new state.  Every dot product keeps its k-ascending
//          order: bit-identical results.
//   wave 2 "cost", wave 3 "control" (below).
// What keeps synchronisation and everything else off the T-step recurrence:
//   * layer 0 (two k-steps) is computed by BOTH dynamics waves, so only the layers after it are
//     split and a step has NHID-1 activation swaps instead of NHID;
//   * the four waves of a group are always co-resident (one workgroup), so they hand data over
//     through LDS sequence words instead of s_barrier.  The LDS executes a wave's instructions
//     in order: a wave writes its data, then its sequence word; a reader that sees the sequence
//     word in one ds_read sees the data in the ds_read it issued after it.  A swap is then one
//     store + one load round trip when the partner is already there;
//   * wave 0 owns the M tiles that feed the first half of the next layer's k-steps, so it runs
//     those MFMAs while its partner's half is in flight;
//   * wave 3 is the "control" wave: noise (in-kernel MRG32k3a + Box-Muller, or the explicit eps
//     buffer), the perturbed control, its write-back to HBM (before the clamp, Q3), the clamp, and
//     the (u, du) record of the cost wave -- nothing of mppi_controller.cu:136-153 depends on the
//     state, so it runs kRing steps ahead and the dynamics waves touch no global memory at all:
//     per step they read one LDS word (their layer-0 B operand [u0,u1,0,0][g]);
//   * wave 2 is the cost wave, software-pipelined by one step around its two costmap fetches.
// Rings of kRing steps decouple the waves; in steady state only the two dynamics waves wait, and
// only for each other.  Every spin loop draws on a per-wave budget; a wave that exhausts it stops
// waiting and raises the group's fail word, and the cost wave poisons the costs with NaN
// (mppi_device.hpp): a loud failure instead of a hung GPU, whichever of the four waves starved.
// Arithmetic and its order are those of the other kernel forms: results are bit-identical.
// ---------------------------------------------------------------------------------------------
// Diagnostic build only (-DMPPI_STAMPS, tools/quad_stamps.sh): s_memtime stamps of the two dynamics waves of
// workgroup 0, accumulated over the steps t >= 16 into a buffer nothing else reads.  The product build has
// no stamp instruction (MI355X guide, "In-kernel stamps").
#ifdef MPPI_STAMPS
__device__ unsigned long long g_quad_stamps[2][8];
#define QSTAMP(var)                                                                                   \
  do {                                                                                                \
    unsigned long long t__;                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                        \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    var = t__;                                                                                        \
  } while (0)
#else
#define QSTAMP(var) do { } while (0)
#endif

constexpr int kRing = 16;  // steps in flight between the waves (power of two)
constexpr int kCtlChunk = 4;  // steps of U / explicit eps the control wave requests at once

// partner's sequence word, then its data; the control wave's publication count, then the lane's
// layer-0 operand of the next step; the cost wave's consumption count
template <int M2>
__device__ __forceinline__ void quad_poll(uint32_t a_seq, uint32_t a_x, uint32_t a_pub, uint32_t a_b1,
                                           uint32_t a_cd, int &f, f32x4 (&oth)[M2], int &cp, float &b1,
                                           int &cd)
{
  static_assert(M2 == 1 || M2 == 2, "one or two M tiles per dynamics wave");
  if constexpr (M2 == 1) {
    asm volatile(
        "ds_read_b32 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b32 %2, %7\n\tds_read_b32 %3, %8\n\t"
        "ds_read_b32 %4, %9\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(f), "=&v"(oth[0]), "=&v"(cp), "=&v"(b1), "=&v"(cd)
        : "v"(a_seq), "v"(a_x), "v"(a_pub), "v"(a_b1), "v"(a_cd)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b32 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %7 offset:16\n\tds_read_b32 %3, %8\n\t"
        "ds_read_b32 %4, %9\n\tds_read_b32 %5, %10\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(f), "=&v"(oth[0]), "=&v"(oth[M2 - 1]), "=&v"(cp), "=&v"(b1), "=&v"(cd)
        : "v"(a_seq), "v"(a_x), "v"(a_pub), "v"(a_b1), "v"(a_cd)
        : "memory");
  }
  f = __builtin_amdgcn_readfirstlane(f);
  cp = __builtin_amdgcn_readfirstlane(cp);
  cd = __builtin_amdgcn_readfirstlane(cd);
}

// the swap alone: partner's sequence word, then its data (one round trip)
template <int M2>
__device__ __forceinline__ void quad_poll_swap(uint32_t a_seq, uint32_t a_x, int &f, f32x4 (&oth)[M2])
{
  static_assert(M2 == 1 || M2 == 2, "one or two M tiles per dynamics wave");
  if constexpr (M2 == 1) {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(oth[0])
                 : "v"(a_seq), "v"(a_x)
                 : "memory");
  } else {
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f), "=&v"(oth[0]), "=&v"(oth[M2 - 1])
                 : "v"(a_seq), "v"(a_x)
                 : "memory");
  }
  f = __builtin_amdgcn_readfirstlane(f);
}

template <int H, int NHID>
struct QuadShared {
  static constexpr int NX = (H / 32) * 4;  // floats a lane hands to its partner per swap
  float xb[2][2][64][NX];                  // [swap parity][wave][lane][.]
  int xseq[2][64];                         // swaps published by dynamics wave w (written per lane, word 0 is read)
  float rec[kRing][kRolloutsPerWave][4];   // s3..s6 before the update of step t (dynamics wave 0)
  int cost_done[64];                       // steps consumed by the cost wave
  float ctl_b1[kRing][64];                 // layer-0 B operand of k-step 1, [u0c, u1c, 0, 0][g] per rollout
  float ctl_rec[kRing][kRolloutsPerWave][4];  // clamped u0, u1, du0, du1 for the cost wave
  int ctl_pub[64];                         // steps published by the control wave
  int fail[4];                             // word 0: raised by a wave whose waits ran out of budget (mppi_device.hpp)
  int fin[4];                              // word r: wave r is through its T steps
};

template <int H, int NHID, int W>
__device__ __forceinline__ void quad_dynamics(const RolloutArgs &a, QuadShared<H, NHID> &sh)
{
  using N = MfmaNet<H, NHID>;
  constexpr int MT = N::MT, M2 = MT / 2, KSH = N::KSH, KS2 = KSH / 2, NX = M2 * 4, NSW = NHID - 1;
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, g = lane >> 4;
  const int T = a.T;

  float A0[MT * 2], B0[MT * 4], AH[NSW * M2 * KSH], Bh[NSW * NX], AL[KSH], BL[4];
#pragma unroll
  for (int i = 0; i < MT * 2; i++) A0[i] = a.wpack[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < MT * 4; i++) B0[i] = a.wpack[(N::nA + i) * 64 + lane] * kTanhScale;
#pragma unroll
  for (int l = 1; l < NHID; l++)
#pragma unroll
    for (int i = 0; i < M2; i++) {
      const int m = W * M2 + i;
#pragma unroll
      for (int s = 0; s < KSH; s++)
        AH[((l - 1) * M2 + i) * KSH + s] = a.wpack[(N::nA0 + (l - 1) * N::nAH + m * KSH + s) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; r++)
        Bh[(l - 1) * NX + i * 4 + r] = a.wpack[(N::nA + l * MT * 4 + m * 4 + r) * 64 + lane] * kTanhScale;
    }
#pragma unroll
  for (int s = 0; s < KSH; s++) AL[s] = a.wpack[(N::nA0 + NSW * N::nAH + s) * 64 + lane];
#pragma unroll
  for (int r = 0; r < 4; r++) BL[r] = a.wpack[(N::nA + NHID * MT * 4 + r) * 64 + lane];

  const uint32_t a_myseq = lds_addr(&sh.xseq[W][lane]);
  const uint32_t a_seq = lds_addr(&sh.xseq[1 - W][0]);
  const uint32_t a_pub = lds_addr(&sh.ctl_pub[0]);
  const uint32_t a_cd = lds_addr(&sh.cost_done[0]);
  const uint32_t a_xmine = lds_addr(&sh.xb[0][W][lane][0]);
  const uint32_t a_xoth = lds_addr(&sh.xb[0][1 - W][lane][0]);
  const uint32_t a_b1 = lds_addr(&sh.ctl_b1[0][lane]);
  constexpr uint32_t kXbParity = 2 * 64 * NX * 4, kB1Slot = 64 * 4;

  float s3 = a.state[3], s4 = a.state[4], s5 = a.state[5], s6 = a.state[6];
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == W + 1);
  while (lds_peek(a_pub) < 1 && --budget > 0) __builtin_amdgcn_s_sleep(1);
  float b1_next = sh.ctl_b1[0][lane];
  int cd = 0;  // last value seen of the cost wave's consumption counter
  // Two forms of the per-step look at the control wave's and the cost wave's progress.  Classic: read with
  // every poll of the swap (one batch of five LDS reads per poll).  Slim: the swap polls only the partner's
  // sequence word and tile; the three other words are requested once after the swap and used after the
  // output layer (LDS-typed volatile loads: plain ds_read instructions whose completion the compiler tracks).
  // Measured (K=4096, T=100, same box): 6-32-32-4 classic 71.3 us / slim 73.3 us; 6-64-64-4 140.2 / 135.4;
  // 6-32x4-4 128.0 / 121.8 -- the slim form wins where a poll carries more data or there are more swaps.
  constexpr bool kSlimPoll = !(H == 32 && NHID == 2);
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile float __attribute__((address_space(3))) *lds_float_p;
  const lds_int_p p_pub = (lds_int_p)&sh.ctl_pub[0];
  const lds_int_p p_cd = (lds_int_p)&sh.cost_done[0];
  const lds_float_p p_b1 = (lds_float_p)&sh.ctl_b1[0][lane];
  int cp_v = 0, cd_v = 0;
  float b1n_v = 0.0f;
#ifdef MPPI_STAMPS
  unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, qprev = 0, qa[6] = {0, 0, 0, 0, 0, 0};
#endif
  // Steps 0 .. T-2 in full.  The update of the LAST step feeds nothing -- the rollout's cost is the running mean
  // over the states BEFORE the updates of steps 1..T-1 (mppi_controller.cu:160-177; the crash flag of the final
  // state is never read) -- so for t = T-1 only the state record goes out, below the loop, and the network is not
  // evaluated (the loop body itself is untouched: an exit test inside it cost 3.6 us per launch).
  for (int t = 0; t < T - 1; t++) {
    QSTAMP(q0);
    const float b1 = b1_next;  // [u0, u1, 0, 0][g] after the clamp (control wave)
    const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
    if (W == 0) {  // record for the cost wave: the state BEFORE this step's update
      while (cd < t - kRing + 1 && --budget > 0) cd = lds_peek(a_cd);
      sh.rec[t & (kRing - 1)][j][g] = b0;
    }
    // layer 0, all M tiles (both waves)
    float act[MT * 4];
    {
      f32x4 acc0[MT];
#pragma unroll
      for (int m = 0; m < MT; m++) {
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        z = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[m * 2 + 0], b0, z, 0, 0, 0);
        acc0[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[m * 2 + 1], b1, z, 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 v = tanh_bias2(f32x2{acc0[m][r], acc0[m][r + 1]}, f32x2{B0[m * 4 + r], B0[m * 4 + r + 1]});
          act[m * 4 + r] = v.x;
          act[m * 4 + r + 1] = v.y;
        }
    }
    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 acc[M2];
    QSTAMP(q1);  // layer 0 (MFMAs + tanh) done
#pragma unroll
    for (int l = 1; l < NHID; l++) {
      // own M tiles of hidden layer l; after the first swap wave 0 arrives with k-steps 0..KS2-1
      // already accumulated
      if (!(W == 0 && l > 1)) {
#pragma unroll
        for (int i = 0; i < M2; i++) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
#pragma unroll
      for (int s = (W == 0 && l > 1) ? KS2 : 0; s < KSH; s++)
#pragma unroll
        for (int i = 0; i < M2; i++)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(AH[((l - 1) * M2 + i) * KSH + s], act[s], acc[i], 0, 0, 0);
      const int n = t * NSW + l;  // 1-based swap count
      const uint32_t par = (uint32_t)(n & 1) * kXbParity;
      f32x4 own[M2], oth[M2];
#pragma unroll
      for (int i = 0; i < M2; i++) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 v = tanh_bias2(f32x2{acc[i][r], acc[i][r + 1]},
                                     f32x2{Bh[(l - 1) * NX + i * 4 + r], Bh[(l - 1) * NX + i * 4 + r + 1]});
          own[i][r] = v.x;
          own[i][r + 1] = v.y;
        }
        lds_put4(a_xmine + par + 16 * i, own[i]);
      }
      lds_publish(a_myseq, n);
      if (l == 1) QSTAMP(q2);  // own tile of the hidden layer: MFMAs + tanh + store + publish done
      if (W == 0) {
        // k-steps 0..KS2-1 of the next layer read wave 0's own tiles: run them while the partner's half
        // is in flight
#pragma unroll
        for (int m = 0; m < M2; m++)
#pragma unroll
          for (int r = 0; r < 4; r++) act[m * 4 + r] = own[m][r];
        if (l == NHID - 1) {
#pragma unroll
          for (int s = 0; s < KS2; s++) o = __builtin_amdgcn_mfma_f32_16x16x4f32(AL[s], act[s], o, 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < M2; i++) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int s = 0; s < KS2; s++)
#pragma unroll
            for (int i = 0; i < M2; i++)
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(AH[(l * M2 + i) * KSH + s], act[s], acc[i], 0, 0, 0);
        }
      }
      // with every swap: the lane's layer-0 operand of step t+1 and the cost wave's progress (used
      // after the first swap of a step only)
      const uint32_t a_b1n = a_b1 + (uint32_t)((t + 1) & (kRing - 1)) * kB1Slot;
      const int want_ctl = (l == 1) ? min(t + 2, T) : 0;
      if constexpr (kSlimPoll) {
        int f;
        for (;;) {
          quad_poll_swap<M2>(a_seq, a_xoth + par, f, oth);
          if ((f >= n) || --budget <= 0) break;
        }
        if (l == 1) {
          // requested now, used at the end of the step (behind the output layer's MFMAs): the control wave's
          // count, then this lane's layer-0 operand of step t+1 (valid if the count read before it is
          // >= t+2), and the cost wave's progress
          cp_v = *p_pub;
          b1n_v = p_b1[((t + 1) & (kRing - 1)) * 64];
          cd_v = *p_cd;
          QSTAMP(q3);
        }
      } else {
        int f, cp, cdn;
        float b1n;
        for (;;) {
          quad_poll<M2>(a_seq, a_xoth + par, a_pub, a_b1n, a_cd, f, oth, cp, b1n, cdn);
          if (((f >= n) & (cp >= want_ctl)) || --budget <= 0) break;
        }
        if (l == 1) {
          b1_next = b1n;
          cd = cdn;
          QSTAMP(q3);  // (wave 0: early output-layer MFMAs, then) the partner's tile has arrived
        }
      }
      // activation of k-step s = 4m + r: from the wave that owns tile m
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) act[m * 4 + r] = ((m / M2) == W) ? own[m % M2][r] : oth[m % M2][r];
    }
#pragma unroll
    for (int s = (W == 0) ? KS2 : 0; s < KSH; s++) o = __builtin_amdgcn_mfma_f32_16x16x4f32(AL[s], act[s], o, 0, 0, 0);
    s3 = fmaf(o[0] + BL[0], a.dt, s3);  // incrementState, neural_net_model.cu:334-344
    s4 = fmaf(o[1] + BL[1], a.dt, s4);
    s5 = fmaf(o[2] + BL[2], a.dt, s5);
    s6 = fmaf(o[3] + BL[3], a.dt, s6);
    if constexpr (kSlimPoll) {
      __builtin_amdgcn_sched_barrier(0);
      const int want = min(t + 2, T);
      int cp = __builtin_amdgcn_readfirstlane(cp_v);
      while (cp < want && --budget > 0) {  // never in steady state: the control wave runs ahead
        cp = __builtin_amdgcn_readfirstlane(*p_pub);
        b1n_v = p_b1[((t + 1) & (kRing - 1)) * 64];
        cd_v = *p_cd;
      }
      b1_next = b1n_v;
      cd = __builtin_amdgcn_readfirstlane(cd_v);
    }
#ifdef MPPI_STAMPS
    QSTAMP(q4);
    if (t >= 16) { qa[0] += q1 - q0; qa[1] += q2 - q1; qa[2] += q3 - q2; qa[3] += q4 - q3; qa[4] += 1; qa[5] += q0 - qprev; }
    qprev = q4;
#endif
  }
#ifdef MPPI_STAMPS
  if (blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 6; i++) g_quad_stamps[W][i] = qa[i];
#endif
  if (W == 0) {  // the record of step T-1, released to the cost wave by the sequence word it waits for
    const int t = T - 1;
    const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
    while (cd < t - kRing + 1 && --budget > 0) cd = lds_peek(a_cd);
    sh.rec[t & (kRing - 1)][j][g] = b0;
    lds_publish(a_myseq, t * NSW + 1);
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[W]));
}

// One group of 16 rollouts of problem instance `a`: the body of the quad kernel.  `group` is the group's index
// INSIDE its instance (blockIdx.x in the stand-alone kernel and in the batched kernel below, whose blockIdx.y is
// the instance).
template <int H, int NHID, bool AFFINE, bool CTRL>
__device__ __forceinline__ void quad_group(const RolloutArgs &a, QuadShared<H, NHID> &sh, const int group)
{
  static_assert((H / 16) % 2 == 0 && NHID >= 2, "needs two M tiles and a hidden layer to split");
  constexpr int NSW = NHID - 1;
  const int lane = threadIdx.x & 63;
  const int role = threadIdx.x >> 6;  // wave-uniform
  const int j = lane & 15;
  const int k = group * kRolloutsPerWave + j;
  const int K = a.K, T = a.T;
  // sequence words start at 0, the constant rows of the layer-0 operand at 0; the only barrier
  if (role == 0) { sh.xseq[0][lane] = 0; sh.xseq[1][lane] = 0; sh.cost_done[lane] = 0; sh.ctl_pub[lane] = 0; sh.fail[lane & 3] = 0; sh.fin[lane & 3] = 0; }
  if (role == 3)
    for (int q = 0; q < kRing; q++) sh.ctl_b1[q][lane] = 0.0f;
  __syncthreads();

  if (role == 3) {
    // -------------------------------- control wave --------------------------------
    const bool inl = a.inline_noise != 0;
    const bool active = lane < kRolloutsPerWave;
    Mrg gsta{0, 0, 0, 0, 0, 0};
    if (active && inl) {
      gsta.s10 = a.rng_in[k]; gsta.s11 = a.rng_in[K + k]; gsta.s12 = a.rng_in[2 * K + k];
      gsta.s20 = a.rng_in[3 * K + k]; gsta.s21 = a.rng_in[4 * K + k]; gsta.s22 = a.rng_in[5 * K + k];
    }
    float2 *const noise = reinterpret_cast<float2 *>(a.noise);
    const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
    const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
    const bool pure_noise_k = (k >= a.k99);  // :141
    const uint32_t a_seq0 = lds_addr(&sh.xseq[0][0]), a_seq1 = lds_addr(&sh.xseq[1][0]);
    const uint32_t a_cd = lds_addr(&sh.cost_done[0]);
    const uint32_t a_mypub = lds_addr(&sh.ctl_pub[lane]);
    int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == 4);
    int seen_x = 0, seen_c = 0;  // swaps published by both dynamics waves / steps consumed by the cost wave
    for (int t0 = 0; t0 < T; t0 += kCtlChunk) {
      // the chunk's nominal controls and (explicit noise) eps are requested together
      float2 Uq[kCtlChunk], eq[kCtlChunk];
#pragma unroll
      for (int q = 0; q < kCtlChunk; q++) {
        const int tq = min(t0 + q, T - 1);
        Uq[q] = Useq[tq];
        eq[q] = (active && !inl) ? noise[(size_t)tq * K + k] : make_float2(0.0f, 0.0f);
      }
#pragma unroll
      for (int q = 0; q < kCtlChunk; q++) {
        const int t = t0 + q;
        if (t < T) {
          // slot t % kRing held step t - kRing: the dynamics waves read it during step t - kRing - 1
          // (done once both published the first swap of step t - kRing), the cost wave in step t - kRing
          const int need_x = (t >= kRing) ? (t - kRing) * NSW + 1 : 0;
          const int need_c = t - kRing + 1;
          while ((seen_x < need_x || seen_c < need_c) && --budget > 0) {
            seen_x = min(lds_peek(a_seq0), lds_peek(a_seq1));
            seen_c = lds_peek(a_cd);
            if (seen_x < need_x || seen_c < need_c) __builtin_amdgcn_s_sleep(2);
          }
          if (active) {
            const float2 e = inl ? noise_pair(gsta) : eq[q];
            // control perturbation, mppi_controller.cu:136-153
            const bool nf = noise_free_k | (t < a.opt_delay);
            const float n0 = e.x * a.nu[0], n1 = e.y * a.nu[1];
            const float du0 = nf ? 0.0f : n0, du1 = nf ? 0.0f : n1;
            float u0 = nf ? Uq[q].x : (pure_noise_k ? n0 : Uq[q].x + n0);
            float u1 = nf ? Uq[q].y : (pure_noise_k ? n1 : Uq[q].y + n1);
            noise[(size_t)t * K + k] = make_float2(u0, u1);  // before the clamp (Q3)
            u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
            u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
            const int slot = t & (kRing - 1);
            sh.ctl_b1[slot][lane] = u0;
            sh.ctl_b1[slot][kRolloutsPerWave + lane] = u1;
            *reinterpret_cast<float4 *>(&sh.ctl_rec[slot][lane][0]) = make_float4(u0, u1, du0, du1);
          }
          lds_publish(a_mypub, t + 1);
        }
      }
    }
    if (active && inl) {
      a.rng_out[k] = gsta.s10; a.rng_out[K + k] = gsta.s11; a.rng_out[2 * K + k] = gsta.s12;
      a.rng_out[3 * K + k] = gsta.s20; a.rng_out[4 * K + k] = gsta.s21; a.rng_out[5 * K + k] = gsta.s22;
    }
    spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[3]));
  } else if (role == 2) {
    // -------------------------------- cost wave --------------------------------
    // Software-pipelined by one step: the costmap texels of step t are requested in iteration t and
    // consumed in iteration t+1, so their latency never stalls the consumption of the rings.
    const uint32_t a_seq0 = lds_addr(&sh.xseq[0][0]);
    const uint32_t a_mydone = lds_addr(&sh.cost_done[lane]);
    float x = a.state[0], y = a.state[1], yaw = a.state[2];
    int crash = 0, budget = spin_budget_init(a.spin_budget, T, a.fault_wave == 3), seen = 0;
    float J = 0.0f;
    float tf_p = 0.0f, tb_p = 0.0f;
    CostTerms ct_p{0.0f, 0.0f, 0.0f};
    int rc_p = 0;
    double rt_p = 0.0;
    for (int t = 0; t <= T; t++) {
      float tf = 0.0f, tb = 0.0f;
      CostTerms ct{0.0f, 0.0f, 0.0f};
      int rc = 0;
      double rt = 0.0;
      if (t < T) {
        rt = a.inv_t[t];
        // rec(t) is written before wave 0 publishes the first swap of step t; ctl(t) was published
        // before the dynamics waves could start step t
        const int need = t * NSW + 1;
        while (seen < need && --budget > 0) {
          seen = lds_peek(a_seq0);
          if (seen < need) __builtin_amdgcn_s_sleep(1);
        }
        const float4 r0 = *reinterpret_cast<const float4 *>(&sh.rec[t & (kRing - 1)][j][0]);      // s3 s4 s5 s6
        const float4 r1 = *reinterpret_cast<const float4 *>(&sh.ctl_rec[t & (kRing - 1)][j][0]);  // u0 u1 du0 du1
        lds_publish(a_mydone, t + 1);  // executes after the two reads (the LDS runs a wave's instructions in order)
        rc = (int)((t > 0) & (fabsf(r0.x) >= kRollCrash));  // getCrash of update t-1
        float spsi, cpsi;
        sincos_fast(yaw, spsi, cpsi);
        const float st[3] = {x, y, yaw};
        track_fetch<AFFINE>(a.cost, st, cpsi, spsi, tf, tb);
        cost_terms_a<CTRL>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
        // computeKinematics + incrementState for x, y, yaw (neural_net_model.cu:346-355, 334-344)
        const float sd0 = fmaf(cpsi, r0.y, -(spsi * r0.z));
        const float sd1 = fmaf(spsi, r0.y, cpsi * r0.z);
        const float sd2 = a.negate_yaw_der ? -r0.w : r0.w;
        x = fmaf(sd0, a.dt, x);
        y = fmaf(sd1, a.dt, y);
        yaw = fmaf(sd2, a.dt, yaw);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t > 0) {  // finish step t-1: running mean over 1..T-1 (Q5); the t = 0 evaluation is discarded
        const int tp = t - 1;
        crash |= rc_p;
        int crash_new = crash;
        const float c = cost_terms_b(a.cost, ct_p, tf_p, tb_p, crash_new);
        const float Jn = running_mean(J, c, tp, rt_p);
        J = (tp > 0) ? Jn : J;
        crash = (tp > 0) ? crash_new : crash;
      }
      tf_p = tf; tb_p = tb; ct_p = ct; rc_p = rc; rt_p = rt;
    }
    // a hand-over that never arrived, in ANY wave of the group: poison, do not hang (mppi_device.hpp).
    // The other three raise the fail word before their finished word; the kernel cannot end before they do.
    {
      const uint32_t a_f0 = lds_addr(&sh.fin[0]), a_f1 = lds_addr(&sh.fin[1]), a_f3 = lds_addr(&sh.fin[3]);
      while ((lds_peek(a_f0) & lds_peek(a_f1) & lds_peek(a_f3)) == 0 && --budget > 0) __builtin_amdgcn_s_sleep(1);
      if (budget <= 0 || lds_peek(lds_addr(&sh.fail[0])) != 0) J = __builtin_nanf("");
    }
    a.costs[k] = J + 0.0f;
  } else if (role == 0) {
    quad_dynamics<H, NHID, 0>(a, sh);
  } else {
    quad_dynamics<H, NHID, 1>(a, sh);
  }
}

template <int H, int NHID, bool AFFINE, bool CTRL>
__global__ __launch_bounds__(256) void rollout_quad_kernel(const RolloutArgs a)
{
  __shared__ __attribute__((aligned(16))) QuadShared<H, NHID> sh;
  quad_group<H, NHID, AFFINE, CTRL>(a, sh, (int)blockIdx.x);
}

// Several independent MPPI instances in ONE launch (mppi_compute_control_batch: the actual-state and the
// predicted-state controller of runControlLoop, run_control_loop.cuh:218-219, K = 1920 each -- 120 + 120 groups on
// 256 CUs): workgroup (x, y) runs group x of instance y with that instance's own argument block (state, U,
// noise / generator states, costmap, cost parameters).  The per-group code is quad_group, so every instance's
// results equal a stand-alone launch bit for bit.
template <int H, int NHID, bool AFFINE, bool CTRL, int NB>
__global__ __launch_bounds__(256) void rollout_quad_batch_kernel(const QuadBatchArgsT<NB> b)
{
  __shared__ __attribute__((aligned(16))) QuadShared<H, NHID> sh;
  // grid (groups of the largest instance, instances): the instance from the workgroup's own index, its argument block at a
  // compile-time position of the segment (MPPI_BATCH_DISPATCH, mppi_device.hpp)
#define MPPI_QUAD_BODY(A)                                          \
  do {                                                             \
    if ((int)blockIdx.x >= (A).K / kRolloutsPerWave) return;       \
    quad_group<H, NHID, AFFINE, CTRL>((A), sh, (int)blockIdx.x);   \
  } while (0)
  MPPI_BATCH_DISPATCH(NB, b, MPPI_QUAD_BODY);
#undef MPPI_QUAD_BODY
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
  if (block_threads == 512) {  // quad form: two dynamics waves + cost wave + control wave per 16 rollouts
    const dim3 grid(a.K / kRolloutsPerWave), block(256);
    if (affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_quad_kernel<H, NHID, true, false>), grid, block, 0, stream, a);
    else if (affine && ctrl) MPPI_LAUNCH_ROLLOUT((rollout_quad_kernel<H, NHID, true, true>), grid, block, 0, stream, a);
    else if (!affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_quad_kernel<H, NHID, false, false>), grid, block, 0, stream, a);
    else MPPI_LAUNCH_ROLLOUT((rollout_quad_kernel<H, NHID, false, true>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  const int waves = a.K / kRolloutsPerWave;
  const int wpb = block_threads / 64;
  const dim3 grid((waves + wpb - 1) / wpb), block(block_threads);
  if (affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_mfma_kernel<H, NHID, true, false>), grid, block, 0, stream, a);
  else if (affine && ctrl) MPPI_LAUNCH_ROLLOUT((rollout_mfma_kernel<H, NHID, true, true>), grid, block, 0, stream, a);
  else if (!affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_mfma_kernel<H, NHID, false, false>), grid, block, 0, stream, a);
  else MPPI_LAUNCH_ROLLOUT((rollout_mfma_kernel<H, NHID, false, true>), grid, block, 0, stream, a);
  return hipGetLastError();
}

template <int H, int NHID, int NB>
static void launch_quad_batch_nb(const QuadBatchArgsT<NB> &b, dim3 grid, bool affine, bool ctrl, hipStream_t stream)
{
  const dim3 block(256);
  if (affine && !ctrl) hipLaunchKernelGGL((rollout_quad_batch_kernel<H, NHID, true, false, NB>), grid, block, 0, stream, b);
  else if (affine && ctrl) hipLaunchKernelGGL((rollout_quad_batch_kernel<H, NHID, true, true, NB>), grid, block, 0, stream, b);
  else if (!affine && !ctrl) hipLaunchKernelGGL((rollout_quad_batch_kernel<H, NHID, false, false, NB>), grid, block, 0, stream, b);
  else hipLaunchKernelGGL((rollout_quad_batch_kernel<H, NHID, false, true, NB>), grid, block, 0, stream, b);
}
template <int H, int NHID>
static hipError_t launch_quad_batch_t(const QuadBatchArgs &b, bool affine, bool ctrl, hipStream_t stream)
{
  int gmax = 0;
  for (int i = 0; i < b.n; i++) gmax = b.inst[i].K / kRolloutsPerWave > gmax ? b.inst[i].K / kRolloutsPerWave : gmax;
  const dim3 grid(gmax, b.n);
  if (b.n <= 2) launch_quad_batch_nb<H, NHID, 2>(batch_args_prefix<2>(b), grid, affine, ctrl, stream);  // (mppi_device.hpp)
  else launch_quad_batch_nb<H, NHID, 4>(b, grid, affine, ctrl, stream);
  return hipGetLastError();
}

hipError_t launch_rollout_quad_batch(int hidden, int n_hidden, const QuadBatchArgs &b, hipStream_t stream)
{
  if (b.n < 1 || b.n > kMaxBatch) return hipErrorInvalidValue;
  bool affine = true, ctrl = false;
  for (int i = 0; i < b.n; i++) {
    affine = affine && b.inst[i].cost.affine != 0;
    ctrl = ctrl || b.inst[i].cost.need_control_cost != 0;
  }
  if (hidden == 32 && n_hidden == 2) return launch_quad_batch_t<32, 2>(b, affine, ctrl, stream);
  if (hidden == 64 && n_hidden == 2) return launch_quad_batch_t<64, 2>(b, affine, ctrl, stream);
  if (hidden == 32 && n_hidden == 4) return launch_quad_batch_t<32, 4>(b, affine, ctrl, stream);
  if (hidden == 64 && n_hidden == 4) return launch_quad_batch_t<64, 4>(b, affine, ctrl, stream);
  return hipErrorInvalidValue;
}

bool mfma_variant_supported(int hidden, int n_hidden)
{
  return (hidden == 32 || hidden == 64) && (n_hidden == 2 || n_hidden == 4);
}

int mfma_pack_floats_per_lane(int hidden, int n_hidden)
{
  const int MT = hidden / 16, KSH = hidden / 4;
  return MT * 2 + (n_hidden - 1) * MT * KSH + KSH + n_hidden * MT * 4 + 4 + (4 * KSH + 1);  // + the tree image (mfma_net.hpp: MfmaTree)
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

#ifdef MPPI_STAMPS
extern "C" int mppi_debug_read_quad_stamps(unsigned long long *out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mppi::g_quad_stamps), sizeof(unsigned long long) * 16);
}
#endif
