// rollout_oct.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) for 64-wide networks at K <= 16 x #CUs:
// EIGHT wavefronts per 16 rollouts -- four dynamics waves, one per 16-row M tile of the layers, and the four
// riders of group_roles.hpp (pose -> cost, noise -> control).
//
// Why: at one group per CU the T-step recurrence is a latency chain, and for a 64-wide net its length is the
// MFMA issue time (32.6 cycles each).  The quad form gives a dynamics wave half of a hidden
// layer (two tiles: 32 MFMAs + 8 tanh per layer); here a wave owns ONE tile (16 MFMAs + 4 tanh), at the same
// single hand-over latency per layer -- the swap becomes all-to-all (a lane stores its 16 B and loads the 16 B
// of the same lane of the three other waves: with the row permutation of mfma_net.hpp, lane l of every wave
// holds k-slot g = l >> 4 of its tile's four k-steps).  Layer 0 is split as well: 2 MFMAs + 4 tanh per wave and
// a swap instead of 8 + 16 (measured, K=4096, T=100: 6-64-64-4 126 vs 146 us, 6-64x4-4 185 vs 198 us).  The
// output layer is one M tile, a serial chain: all four compute it, so all four hold the new state.  Every dot
// product keeps its k-ascending order: bit-identical to the other forms.
//
// The four dynamics waves take the four SIMDs of the CU, so everything else rides along in the issue slots they
// leave free (the waits of the swaps).  With two riders (cost wave, control wave with the generator) the
// control wave could not keep up with a 6-64-64-4 step: rollout 125 us; with the work cut into four stages, one
// per SIMD, 107 us (101 us when eps comes from the generator kernel: the dynamics chain alone).
//
// Hand-overs: LDS sequence words, no barrier in the T loop (rollout_mfma.hip, quad form).  Two parities of the
// swap buffer suffice: a wave publishes swap n+1 only after it has loaded every partner's swap n, so when a
// wave writes swap n+2 into the slot of swap n, all partners have published n+1, i.e. are done reading n.
#include "mfma_net.hpp"
#include "group_roles.hpp"
#include "mppi_kernels.hpp"

namespace mppi {

template <int H, int NHID>
struct OctShared {
  static constexpr int NW = 4;
  static constexpr int NSW = NHID;  // swaps per step: layer 0 and the NHID-1 hidden -> hidden layers
  static constexpr int kR = 16;           // rollouts per group
  static constexpr bool kRecByAll = false;  // dynamics wave 0 writes the state records (all four hold the state)
  float xb[2][NW][64][4];                   // [swap parity][wave][lane][.]: a wave's tile of activations
  int xseq[NW][64];                         // swaps published by dynamics wave w
  float rec[kGRing][kRolloutsPerWave][4];
  int cost_done[64];
  float ctl_b1[kGRing][64];
  float ctl_rec[kGRing][kRolloutsPerWave][4];
  int ctl_pub[64];
  float tex[kGRing][kRolloutsPerWave][2];   // pose wave -> cost wave
  int pose_pub[64];
  float eps[kGRing][kRolloutsPerWave][2];   // noise wave -> control wave
  int rng_pub[64];
  int fail[4];
  int fin[8];
};

// the three partners' sequence words and tiles, one round trip
__device__ __forceinline__ void oct_poll(const uint32_t (&a_seq)[3], const uint32_t (&a_x)[3], int &f, f32x4 (&oth)[3])
{
  int f0, f1, f2;
  asm volatile(
      "ds_read_b32 %0, %6\n\tds_read_b128 %3, %9\n\t"
      "ds_read_b32 %1, %7\n\tds_read_b128 %4, %10\n\t"
      "ds_read_b32 %2, %8\n\tds_read_b128 %5, %11\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(oth[0]), "=&v"(oth[1]), "=&v"(oth[2])
      : "v"(a_seq[0]), "v"(a_seq[1]), "v"(a_seq[2]), "v"(a_x[0]), "v"(a_x[1]), "v"(a_x[2])
      : "memory");
  f = __builtin_amdgcn_readfirstlane(min(min(f0, f1), f2));
}

template <int H, int NHID, int W>
__device__ __forceinline__ void oct_dynamics(const RolloutArgs &a, OctShared<H, NHID> &sh)
{
  using N = MfmaNet<H, NHID>;
  constexpr int MT = N::MT, KSH = N::KSH, NSW = NHID;
  static_assert(MT == 4, "one M tile per dynamics wave: 64-wide hidden layers");
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, g = lane >> 4;
  const int T = a.T;

  // A-operand and bias slices of this wave: layer 0 (own tile or all), own tile of every hidden layer, output layer
  float A0[2], B0[4], AH[(NHID - 1) * KSH + 1], Bh[(NHID - 1) * 4 + 1], AL[KSH], BL[4];
  A0[0] = a.wpack[(W * 2 + 0) * 64 + lane];
  A0[1] = a.wpack[(W * 2 + 1) * 64 + lane];
#pragma unroll
  for (int r = 0; r < 4; r++) B0[r] = a.wpack[(N::nA + W * 4 + r) * 64 + lane] * kTanhScale;
#pragma unroll
  for (int l = 1; l < NHID; l++) {
#pragma unroll
    for (int s = 0; s < KSH; s++)
      AH[(l - 1) * KSH + s] = a.wpack[(N::nA0 + (l - 1) * N::nAH + W * KSH + s) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; r++) Bh[(l - 1) * 4 + r] = a.wpack[(N::nA + l * MT * 4 + W * 4 + r) * 64 + lane] * kTanhScale;
  }
#pragma unroll
  for (int s = 0; s < KSH; s++) AL[s] = a.wpack[(N::nA0 + (NHID - 1) * N::nAH + s) * 64 + lane];
#pragma unroll
  for (int r = 0; r < 4; r++) BL[r] = a.wpack[(N::nA + NHID * MT * 4 + r) * 64 + lane];

  const uint32_t a_myseq = lds_addr(&sh.xseq[W][lane]);
  const uint32_t a_xmine = lds_addr(&sh.xb[0][W][lane][0]);
  uint32_t a_seq[3], a_xo[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    a_seq[i] = lds_addr(&sh.xseq[(W + 1 + i) & 3][0]);
    a_xo[i] = lds_addr(&sh.xb[0][(W + 1 + i) & 3][lane][0]);
  }
  const uint32_t a_pub = lds_addr(&sh.ctl_pub[0]);
  const uint32_t a_cd = lds_addr(&sh.cost_done[0]);
  constexpr uint32_t kXbParity = 4 * 64 * 4 * 4;

  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile float __attribute__((address_space(3))) *lds_float_p;
  const lds_int_p p_pub = (lds_int_p)&sh.ctl_pub[0];
  const lds_int_p p_cd = (lds_int_p)&sh.cost_done[0];
  const lds_float_p p_b1 = (lds_float_p)&sh.ctl_b1[0][lane];

  float s3 = a.state[3], s4 = a.state[4], s5 = a.state[5], s6 = a.state[6];
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == W + 1);
  while (lds_peek(a_pub) < 1 && --budget > 0) __builtin_amdgcn_s_sleep(1);
  float b1_next = sh.ctl_b1[0][lane];
  // pinned: the wait for this LDS read sits here, not at the first use of b1 inside the loop, where it would also
  // wait, every step, for the state record stored just before it (rollout_multi.hip, multi_dynamics)
  asm volatile("" : "+v"(b1_next));
  int cp_v = 0, cd_v = 0;
  float b1n_v = 0.0f;

  // one swap: store the own tile, publish, load the three others; act[] = the 16 k-step operands of the next layer
  auto swap = [&](int n, const f32x4 &own, float (&act)[MT * 4], bool first_of_step, int t) __attribute__((always_inline)) {
    const uint32_t par = (uint32_t)(n & 1) * kXbParity;
    lds_put4(a_xmine + par, own);
    lds_publish(a_myseq, n);
    const uint32_t ax[3] = {a_xo[0] + par, a_xo[1] + par, a_xo[2] + par};
    f32x4 oth[3];
    int f;
    for (;;) {
      oct_poll(a_seq, ax, f, oth);
      if (f >= n || --budget <= 0) break;
    }
    if (first_of_step) {
      // requested now, used at the end of the step (behind the output layer's MFMAs): the control wave's count,
      // then this lane's layer-0 operand of step t+1 (valid if the count read before it is >= t+2), and the
      // cost wave's progress
      cp_v = *p_pub;
      b1n_v = p_b1[((t + 1) & (kGRing - 1)) * 64];
      cd_v = *p_cd;
    }
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) act[m * 4 + r] = (m == W) ? own[r] : oth[(m - W - 1) & 3][r];
  };

  // Steps 0 .. T-2 in full; of step T-1 only the state record goes out (its update feeds nothing: the cost is
  // the running mean over the states BEFORE the updates of steps 1..T-1, mppi_controller.cu:160-177)
  for (int t = 0; t < T - 1; t++) {
    const float b1 = b1_next;  // [u0, u1, 0, 0][g] after the clamp (control wave)
    const float b0 = row_sel(g, s3, s4, s5, s6);
    float act[MT * 4];
    int n = t * NSW;  // swaps published before this step
    {
      // k-step 0 needs the state only: its MFMA goes first, wave 0's record store is issued in its shadow
      f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
      z = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[0], b0, z, 0, 0, 0);
      // record for the cost wave: the state BEFORE this step's update (its ring slot was checked at the end of
      // the previous step)
      if (W == 0) sh.rec[t & (kGRing - 1)][j][g] = b0;
      z = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[1], b1, z, 0, 0, 0);
      f32x4 own;
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2 v = tanh_bias2(f32x2{z[r], z[r + 1]}, f32x2{B0[r], B0[r + 1]});
        own[r] = v.x;
        own[r + 1] = v.y;
      }
      swap(++n, own, act, true, t);
    }
#pragma unroll
    for (int l = 1; l < NHID; l++) {
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < KSH; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(AH[(l - 1) * KSH + s], act[s], acc, 0, 0, 0);
      f32x4 own;
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2 v = tanh_bias2(f32x2{acc[r], acc[r + 1]}, f32x2{Bh[(l - 1) * 4 + r], Bh[(l - 1) * 4 + r + 1]});
        own[r] = v.x;
        own[r + 1] = v.y;
      }
      swap(++n, own, act, false, t);
    }
    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < KSH; s++) o = __builtin_amdgcn_mfma_f32_16x16x4f32(AL[s], act[s], o, 0, 0, 0);
    s3 = fmaf(o[0] + BL[0], a.dt, s3);  // incrementState, neural_net_model.cu:334-344
    s4 = fmaf(o[1] + BL[1], a.dt, s4);
    s5 = fmaf(o[2] + BL[2], a.dt, s5);
    s6 = fmaf(o[3] + BL[3], a.dt, s6);
    __builtin_amdgcn_sched_barrier(0);
    // ONE scalar test per step (a second one at the top of wave 0's step costs the whole group ~100 cycles per
    // step, tools/ub/dyn_step_ub.hip): step t+1 may start when the control wave has published it (never late in
    // steady state: it runs ahead) and -- wave 0 -- the ring slot of its state record is free: that slot held
    // step t+1 - kGRing, consumed once cost_done >= t+2 - kGRing
    const int want = t + 2, want_cd = (W == 0) ? t + 2 - kGRing : -(1 << 30);
    int cp = __builtin_amdgcn_readfirstlane(cp_v), cd = __builtin_amdgcn_readfirstlane(cd_v);
    while (((cp < want) | (cd < want_cd)) && --budget > 0) {
      cp = __builtin_amdgcn_readfirstlane(*p_pub);
      b1n_v = p_b1[((t + 1) & (kGRing - 1)) * 64];
      cd = __builtin_amdgcn_readfirstlane(*p_cd);
    }
    b1_next = b1n_v;
  }
  if (W == 0) {  // the record of step T-1, released to the cost wave by the sequence word it waits for
    const int t = T - 1;
    const float b0 = row_sel(g, s3, s4, s5, s6);
    sh.rec[t & (kGRing - 1)][j][g] = b0;
    lds_publish(a_myseq, t * NSW + 1);
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[W]));
}

template <int H, int NHID, bool AFFINE, bool CTRL>
__global__ __launch_bounds__(512) void rollout_oct_kernel(const RolloutArgs a)
{
  using SH = OctShared<H, NHID>;
  using R = GroupRoles<SH>;
  __shared__ __attribute__((aligned(16))) SH sh;
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // sequence words start at 0, the constant rows of the layer-0 operand at 0; the only barrier
  if (role == 0) {
#pragma unroll
    for (int w = 0; w < 4; w++) sh.xseq[w][lane] = 0;
    sh.cost_done[lane] = 0;
    sh.ctl_pub[lane] = 0;
    sh.pose_pub[lane] = 0;
    sh.rng_pub[lane] = 0;
    sh.fail[lane & 3] = 0;
    sh.fin[lane & 7] = 0;
  }
  if (role == R::kCtl)
    for (int q = 0; q < kGRing; q++) sh.ctl_b1[q][lane] = 0.0f;
  __syncthreads();

  if (role == 0) oct_dynamics<H, NHID, 0>(a, sh);
  else if (role == 1) oct_dynamics<H, NHID, 1>(a, sh);
  else if (role == 2) oct_dynamics<H, NHID, 2>(a, sh);
  else if (role == 3) oct_dynamics<H, NHID, 3>(a, sh);
  else if (role == R::kCost) group_cost_wave<SH, CTRL>(a, sh);
  else if (role == R::kCtl) group_control_wave(a, sh);
  else if (role == R::kPose) group_pose_wave<SH, AFFINE>(a, sh);
  else group_rng_wave(a, sh);
}

template <int H, int NHID>
static hipError_t launch_oct_t(const RolloutArgs &a, hipStream_t stream)
{
  const bool affine = a.cost.affine != 0, ctrl = a.cost.need_control_cost != 0;
  const dim3 grid(a.K / kRolloutsPerWave), block(512);
  if (affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_oct_kernel<H, NHID, true, false>), grid, block, 0, stream, a);
  else if (affine && ctrl) MPPI_LAUNCH_ROLLOUT((rollout_oct_kernel<H, NHID, true, true>), grid, block, 0, stream, a);
  else if (!affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_oct_kernel<H, NHID, false, false>), grid, block, 0, stream, a);
  else MPPI_LAUNCH_ROLLOUT((rollout_oct_kernel<H, NHID, false, true>), grid, block, 0, stream, a);
  return hipGetLastError();
}

bool oct_variant_supported(int hidden, int n_hidden)
{
  return hidden == 64 && (n_hidden == 2 || n_hidden == 4);
}

hipError_t launch_rollout_oct(int hidden, int n_hidden, const RolloutArgs &a, hipStream_t stream)
{
  if (a.K % kRolloutsPerWave != 0 || hidden != 64) return hipErrorInvalidValue;
  if (n_hidden == 2) return launch_oct_t<64, 2>(a, stream);
  if (n_hidden == 4) return launch_oct_t<64, 4>(a, stream);
  return hipErrorInvalidValue;
}

}  // namespace mppi
