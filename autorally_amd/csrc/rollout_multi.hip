// rollout_multi.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) for gfx950, the form for K beyond
// the latency regime: ND "dynamics" wavefronts (16 rollouts each, the whole network on the matrix
// instruction, mfma_net.hpp) + ONE cost wavefront (two at ND = 4: pose | cost) + ONE control wavefront per
// workgroup of 16 ND rollouts.
//
// Why: in the single-wave form (rollout_mfma_kernel) every lane of a 16-rollout wave carries the scalar
// work of its rollout -- controls, sin/cos, two costmap fetches, MPPICosts::computeCost, the f64 running
// mean -- 4x redundantly (4 lanes per rollout), and the f32 matrix instruction shares the vector datapath,
// so that work ADDS to the network's cycles: ~1 700 of ~5 750 cycles per step at 6-64-64-4.  Here
//   * a dynamics wave does the network and the Euler update of [roll, u_x, u_y, yaw_mder] only; per step
//     it reads one LDS word (its layer-0 operand [u0, u1, 0, 0][g], prefetched a step ahead) and writes one
//     (the state record of the cost wave); it touches no global memory inside the T loop;
//   * the cost wave serves all 16 ND rollouts with ONE LANE PER ROLLOUT: x, y, yaw kinematics, sin/cos,
//     costmap fetches, computeCost, running mean, crash flags -- software-pipelined by one step around the
//     fetches, exactly the cost wave of the quad kernel with 64 useful lanes instead of 16;
//   * the control wave, one lane per rollout: eps (explicit buffer, requested 4 steps ahead, or the
//     in-kernel MRG32k3a), the perturbed control, its write-back before the clamp (Q3), the clamp, the
//     records of the other waves; it runs up to kRing steps ahead.
// Rings of kRing steps and LDS sequence words decouple the waves (mppi_device.hpp): no barrier in the T
// loop, nobody waits in steady state.  Same arithmetic in the same order as the other forms: bit-identical.
// The waves of one workgroup are spread over the four SIMDs of a CU by the dispatcher, so with ND = 4 and
// one workgroup per CU every SIMD runs exactly one dynamics wave (K = 16384: 256 workgroups).
#include "mfma_net.hpp"
#include "noise_device.hpp"
#include "mppi_kernels.hpp"

namespace mppi {

constexpr int kMRing = 16;    // steps in flight between the waves (power of two)
constexpr int kMCtlChunk = 4;  // steps of U / explicit eps the control wave requests at once

template <int ND, bool SPLIT_ = (ND == 4)>
struct MultiShared {
  static constexpr bool SPLIT = SPLIT_;  // cost work on three wavefronts (pose | fetch | cost), see the kernel
  static constexpr int NR = 16 * ND;       // rollouts per workgroup
  float rec[kMRing][NR][4];                // s3..s6 before the update of step t (dynamics waves)
  float ctl_rec[kMRing][NR][4];            // clamped u0, u1, du0, du1 (control wave -> cost wave)
  float ctl_b1[kMRing][ND][64];            // layer-0 operand of k-step 1, [u0c, u1c, 0, 0][g] in lane order
  int dyn_pub[4][64];                      // steps published by dynamics wave w (written per lane, word 0 read)
  int cost_done[64];                       // steps consumed by the cost wave
  int ctl_pub[64];                         // steps published by the control wave
  float pts[SPLIT ? kMRing : 1][SPLIT ? NR : 1][4];  // x, y, cos psi, sin psi of step t (pose wave -> fetch wave)
  int pose_pub[64];                        // steps the pose wave has published
  float tex[SPLIT ? kMRing : 1][SPLIT ? NR : 1][2];  // costmap texels (front, back) of step t (fetch wave -> cost wave)
  int fetch_pub[64];                       // steps whose texels the fetch wave has published
  int fail[4];                             // word 0: raised by a wave whose waits ran out of budget
  int fin[8];                              // word r: wave r is through its T steps
  float gstate[8];                         // gated launch: the vehicle state the control wave took from the gate block,
  int gate_open[8];                        // then 1 here (group_gate_wait, mppi_device.hpp)
};

__device__ __forceinline__ void lds_put1(uint32_t addr, float v)
{
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// smallest of the ND dynamics waves' publication counts (lane l reads the word of wave l % ND)
template <int ND>
__device__ __forceinline__ int dyn_pub_min(uint32_t a_lane_word)
{
  const int v = lds_peek_lanes(a_lane_word);
  int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
  for (int w = 1; w < ND; w++) m = min(m, __builtin_amdgcn_readlane(v, w));
  return m;
}

// Diagnostic build only (-DMPPI_STAMPS): s_memtime stamps of dynamics wave 0 of workgroup 0 (rollout_mfma.hip).
#ifdef MPPI_STAMPS
__device__ unsigned long long g_multi_stamps[8];
#define MSTAMP(var)                                                              \
  do {                                                                           \
    unsigned long long t__;                                                      \
    __builtin_amdgcn_sched_barrier(0);                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");   \
    __builtin_amdgcn_sched_barrier(0);                                           \
    var = t__;                                                                   \
  } while (0)
#else
#define MSTAMP(var) do { } while (0)
#endif

// TREE: the output layer as a butterfly over the four lanes of a rollout (mfma_net.hpp: nn_last_tree) instead of KSH matrix
// instructions; a lane then carries ONE state component, s[3 + g] -- exactly the layer-0 operand it feeds
template <int H, int NHID, int ND, class SH, bool TREE = false, bool GATED = false>
__device__ __forceinline__ void multi_dynamics(const RolloutArgs &a, SH &sh, const int w)
{
  using N = MfmaNet<H, NHID>;
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, g = lane >> 4;
  const int T = a.T;
  float A[N::nA], Bi[N::nBias];
  load_weights<H, NHID>(a.wpack, lane, A, Bi);
  f32x2 wt[2 * MfmaTree<H, NHID>::KSH];
  float bo = 0.0f;
  if constexpr (TREE) {
    load_tree_weights<H, NHID>(a.wpack, lane, wt, bo);
#pragma unroll
    for (int i = 0; i < 2 * MfmaTree<H, NHID>::KSH; i++) asm volatile("" : "+v"(wt[i]));
  }
  // pinned: the waits for these loads sit here, not (one s_waitcnt vmcnt per first use) inside the T loop
#pragma unroll
  for (int i = 0; i < (TREE ? N::nA - N::nAL : N::nA); i++) asm volatile("" : "+v"(A[i]));
#pragma unroll
  for (int i = 0; i < N::nBias; i++) asm volatile("" : "+v"(Bi[i]));

  const uint32_t a_mypub = lds_addr(&sh.dyn_pub[w][lane]);
  const uint32_t a_rec = lds_addr(&sh.rec[0][16 * w + j][g]);
  constexpr uint32_t kRecSlot = SH::NR * 16, kB1Slot = ND * 64 * 4;
  // LDS-typed volatile pointers: plain ds_read instructions in program order, the compiler keeps track of
  // their completion itself (s_waitcnt at the first use), so a value requested early costs nothing later
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile float __attribute__((address_space(3))) *lds_float_p;
  const lds_int_p p_pub = (lds_int_p)&sh.ctl_pub[0];
  const lds_int_p p_cd = (lds_int_p)&sh.cost_done[0];
  const lds_float_p p_b1 = (lds_float_p)&sh.ctl_b1[0][w][lane];

  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == 1 + w);
  float s3, s4, s5, s6, sv;
  if constexpr (GATED) {  // the state arrives through the gate block: the control wave has put it into LDS
    const uint32_t a_go = lds_addr(&sh.gate_open[0]);
    while (lds_peek(a_go) == 0 && --budget > 0) __builtin_amdgcn_s_sleep(1);
    const volatile float *gs = sh.gstate;
    s3 = gs[3]; s4 = gs[4]; s5 = gs[5]; s6 = gs[6];
    sv = gs[3 + g];
  } else {
    s3 = a.state[3]; s4 = a.state[4]; s5 = a.state[5]; s6 = a.state[6];
    sv = a.state[3 + g];  // TREE: this lane's component
  }
  while (__builtin_amdgcn_readfirstlane(*p_pub) < 1 && --budget > 0) __builtin_amdgcn_s_sleep(1);
  // (s_setprio 3 here -- issue priority over the cost / control wave sharing this wave's SIMD -- changes nothing:
  //  cfg 4 304.4 us without, 305.5 us with; the co-resident wave's instructions cost their issue cycles either way)
  float b1_next = *p_b1;
  // Pinned: the value is "produced" here, so the wait for this LDS read sits here.  Otherwise the compiler's
  // waitcnt pass, which cannot tell the first iteration from the others, puts an s_waitcnt lgkmcnt at the first
  // use of b1 INSIDE the loop, where it waits every step for the record store and the publication issued just
  // before it (LDS operations complete in order; ~20 cycles per step in tools/ub/dyn_step_ub.hip).
  asm volatile("" : "+v"(b1_next));
#ifdef MPPI_STAMPS
  unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0, mprev = 0, ma[6] = {0, 0, 0, 0, 0, 0};
#endif
  // the state record of step t (the state BEFORE its update) for the cost wave, then its publication -- which
  // also says: this wave is done with the control record of step t
  auto record = [&](int t, float b0) __attribute__((always_inline)) {
    lds_put1(a_rec + (uint32_t)(t & (kMRing - 1)) * kRecSlot, b0);
    lds_publish(a_mypub, t + 1);
  };
  // Steps 0 .. T-2 in full; of step T-1 only the record goes out (the last update feeds no cost,
  // mppi_controller.cu:160-177).  ONE scalar branch per step: the ring check of the NEXT step's record joins the
  // test of the control wave's count at the end of the step, and the exit test of the last step is peeled off --
  // a second test and an exit test at the top of the step cost ~210 cycles per step of this chain
  // (tools/ub/dyn_step_ub.hip: 1 935 -> 1 727 cycles at 6-32-32-4, 4 394 -> 4 162 at 6-64-64-4).
  for (int t = 0; t < T - 1; t++) {
    MSTAMP(m0);
    const float b1 = b1_next;  // [u0, u1, 0, 0][g] after the clamp (control wave)
    // k-step 0 of layer 0 needs the state only: its MFMAs go first, and the step's LDS traffic below is issued
    // in their shadow instead of in front of the chain
    const float b0 = TREE ? sv : row_sel(g, s3, s4, s5, s6);
    f32x4 acc[N::MT];
#pragma unroll
    for (int m = 0; m < N::MT; m++)
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 2 + 0], b0, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    record(t, b0);
    // requested now, used after the network: the control wave's count, then this lane's layer-0 operand
    // of step t+1 (valid if the count read before it is >= t+2), and the cost wave's progress
    const int tn = (t + 1) & (kMRing - 1);
    int cp = *p_pub;
    float b1n = p_b1[tn * (kB1Slot / 4)];
    int cd = *p_cd;
    __builtin_amdgcn_sched_barrier(0);  // keep the requests up here: the scheduler would sink them to their use
    MSTAMP(m1);

#pragma unroll
    for (int m = 0; m < N::MT; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * 2 + 1], b1, acc[m], 0, 0, 0);
    nn_hidden<H, NHID>(A, Bi, acc);
    if constexpr (TREE) {
      sv = fmaf(nn_last_tree<H, NHID>(wt, bo, Bi, acc), a.dt, sv);  // incrementState, neural_net_model.cu:334-344
    } else {
      float d[4];
      nn_last<H, NHID>(A, Bi, acc, d);
      s3 = fmaf(d[0], a.dt, s3);  // incrementState, neural_net_model.cu:334-344
      s4 = fmaf(d[1], a.dt, s4);
      s5 = fmaf(d[2], a.dt, s5);
      s6 = fmaf(d[3], a.dt, s6);
    }

    MSTAMP(m2);
    __builtin_amdgcn_sched_barrier(0);  // ... and their first use down here, behind the network
    // step t+1 may start when the control wave has published it (never late in steady state: it runs ahead) and
    // the ring slot of its record is free: that slot held step t+1 - kMRing, consumed once cost_done >= t+2 - kMRing
    const int want = t + 2, want_cd = t + 2 - kMRing;
    cp = __builtin_amdgcn_readfirstlane(cp);
    cd = __builtin_amdgcn_readfirstlane(cd);
    while (((cp < want) | (cd < want_cd)) && --budget > 0) {
      cp = __builtin_amdgcn_readfirstlane(*p_pub);
      b1n = p_b1[tn * (kB1Slot / 4)];
      cd = __builtin_amdgcn_readfirstlane(*p_cd);
    }
    b1_next = b1n;
#ifdef MPPI_STAMPS
    MSTAMP(m3);
    if (t >= 16) { ma[0] += m1 - m0; ma[1] += m2 - m1; ma[2] += m3 - m2; ma[3] += m0 - mprev; ma[4] += 1; }
    mprev = m3;
#endif
  }
  record(T - 1, TREE ? sv : row_sel(g, s3, s4, s5, s6));
#ifdef MPPI_STAMPS
  if (blockIdx.x == 0 && w == 0 && lane == 0)
    for (int i = 0; i < 6; i++) g_multi_stamps[i] = ma[i];
#endif
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[w]));
}

// SPLIT_: the pose and fetch riders of the ND = 4 form (eight waves per workgroup); ND = 2 runs one cost wave.  (Round 3's
// six-wave ND = 4 form, "multi4u", never won a bucket of the selection table and was removed in round 5.)
// GATED (the automatic ND = 4 tree form only): enqueued one solve ahead (a.gate != nullptr), state and nominal sequence from the
// gate block -- the control wave, the first to need host data, waits for the gate (group_gate_wait)
template <int H, int NHID, int ND, bool SPLIT_ = (ND == 4), bool TREE = false, bool GATED = false>
__global__ __launch_bounds__((ND + 2 + (SPLIT_ ? 2 : 0)) * 64, (ND == 4 && !SPLIT_) ? 3 : 1) void rollout_multi_kernel(const RolloutArgs a)
{
  using SH = MultiShared<ND, SPLIT_>;
  constexpr int NR = SH::NR;
  constexpr bool SPLIT = SH::SPLIT;
  // Roles: 0..ND-1 dynamics; then [pose,] cost, control.  With ND = 4 every SIMD of the CU carries a dynamics wave
  // and the other waves ride along on the issue slots its MFMA chain leaves.  That costs the dynamics wave nothing
  // (tools/ub/rider_ub.hip), but a rider whose step is one long dependent chain is stretched -- the one cost wave
  // of the other forms to ~2400 cycles per step -- and through the rings' back-pressure the whole workgroup runs
  // at its pace (6-32-32-4, K=16384: 104.7 us, 82.7 us with the cost arithmetic removed).  So the cost work is a
  // two-stage pipeline on two wavefronts that land on different SIMDs: the POSE wave (x, y, yaw kinematics,
  // sin/cos, the two costmap fetches) hands the texels to the COST wave (computeCost, running mean, crash flags)
  // through one more ring; each stage keeps up.
  // The pose work is cut once more, pose (sin/cos, kinematics) -> FETCH (look-ahead points, texel fetches), the
  // fetch wave being the eighth wave, on the SIMD that had no rider: the f32 MFMA and the vector ALU share one
  // pipeline, so a SIMD's step is max(chain, MFMA + VALU of everything on it), and with ~100 vector instructions
  // per step (f64 range reduction included) the one pose wave pushed its SIMD past the chain of a 32-wide net.
  constexpr int kPose = SPLIT ? ND : -1, kCost = ND + (SPLIT ? 1 : 0), kCtl = kCost + 1, kFetch = SPLIT ? kCtl + 1 : -1;
  __shared__ __attribute__((aligned(16))) SH sh;
  const int lane = threadIdx.x & 63;
  // made uniform for the compiler (budgets and waits stay scalar)
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int K = a.K, T = a.T;
  // sequence words and the constant rows of the layer-0 operand start at 0; the only barrier
  if (role == 0) {
    for (int w = 0; w < 4; w++) sh.dyn_pub[w][lane] = 0;
    sh.cost_done[lane] = 0;
    sh.ctl_pub[lane] = 0;
    sh.pose_pub[lane] = 0;
    sh.fetch_pub[lane] = 0;
    sh.fail[lane & 3] = 0;
    sh.fin[lane & 7] = 0;
    sh.gate_open[lane & 7] = 0;
  }
  if (role == kCtl)
    for (int q = 0; q < kMRing; q++)
      for (int w = 0; w < ND; w++) sh.ctl_b1[q][w][lane] = 0.0f;
  __syncthreads();

  if (role < ND) {
    multi_dynamics<H, NHID, ND, SH, TREE, GATED>(a, sh, role);
  } else if (role == kCtl) {
    // -------------------------------- control wave: one lane per rollout --------------------------------
    const bool inl = a.inline_noise != 0;
    const bool active = lane < NR;
    const int r = active ? lane : NR - 1;
    const int k = blockIdx.x * NR + r;
    Mrg gsta{0, 0, 0, 0, 0, 0};
    if (active && inl) {
      gsta.s10 = a.rng_in[k]; gsta.s11 = a.rng_in[K + k]; gsta.s12 = a.rng_in[2 * K + k];
      gsta.s20 = a.rng_in[3 * K + k]; gsta.s21 = a.rng_in[4 * K + k]; gsta.s22 = a.rng_in[5 * K + k];
    }
    float2 *const noise = reinterpret_cast<float2 *>(a.noise);
    const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
    const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
    const bool pure_noise_k = (k >= a.k99);  // :141
    const uint32_t a_dynw = lds_addr(&sh.dyn_pub[lane & (ND - 1)][0]);
    const uint32_t a_cd = lds_addr(&sh.cost_done[0]);
    const uint32_t a_mypub = lds_addr(&sh.ctl_pub[lane]);
    int shut = 0;
    if constexpr (GATED) shut = group_gate_wait(a, sh);
    int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == kCtl + 1 || shut != 0);
    int seen_d = 0, seen_c = 0;  // steps published by all dynamics waves / consumed by the cost wave
    for (int t0 = 0; t0 < T; t0 += kMCtlChunk) {
      float2 Uq[kMCtlChunk], eq[kMCtlChunk];
#pragma unroll
      for (int q = 0; q < kMCtlChunk; q++) {
        const int tq = min(t0 + q, T - 1);
        if constexpr (GATED) {  // the gate block is host-written memory: system-scope loads
          const unsigned long long ub = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(Useq + tq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          Uq[q] = make_float2(__uint_as_float((unsigned)ub), __uint_as_float((unsigned)(ub >> 32)));
        } else {
          Uq[q] = Useq[tq];
        }
        eq[q] = (active && !inl) ? noise[(size_t)tq * K + k] : make_float2(0.0f, 0.0f);
      }
#pragma unroll
      for (int q = 0; q < kMCtlChunk; q++) {
        const int t = t0 + q;
        if (t < T) {
          // slot t % kMRing held step t - kMRing: a dynamics wave is done with it once it has published
          // step t - kMRing, the cost wave once it has consumed it
          const int need = t - kMRing + 1;
          while ((seen_d < need || seen_c < need) && --budget > 0) {
            seen_d = dyn_pub_min<ND>(a_dynw);
            seen_c = lds_peek(a_cd);
            if (seen_d < need || seen_c < need) __builtin_amdgcn_s_sleep(2);
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
            const int slot = t & (kMRing - 1);
            sh.ctl_b1[slot][r >> 4][r & 15] = u0;
            sh.ctl_b1[slot][r >> 4][16 + (r & 15)] = u1;
            *reinterpret_cast<float4 *>(&sh.ctl_rec[slot][r][0]) = make_float4(u0, u1, du0, du1);
          }
          lds_publish(a_mypub, t + 1);
        }
      }
    }
    if (active && inl) {
      a.rng_out[k] = gsta.s10; a.rng_out[K + k] = gsta.s11; a.rng_out[2 * K + k] = gsta.s12;
      a.rng_out[3 * K + k] = gsta.s20; a.rng_out[4 * K + k] = gsta.s21; a.rng_out[5 * K + k] = gsta.s22;
    }
    spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[kCtl]));
  } else if (SPLIT && role == kPose) {
    // -------------------------------- pose wave (ND = 4): one lane per rollout --------------------------------
    // x, y, yaw of the rollouts: sin/cos and the kinematic update; hands (x, y, cos, sin) of every step on.
    const bool active = lane < NR;
    const int r = active ? lane : NR - 1;
    const uint32_t a_dynw = lds_addr(&sh.dyn_pub[lane & (ND - 1)][0]);
    const uint32_t a_fp = lds_addr(&sh.fetch_pub[0]);
    const uint32_t a_mypub = lds_addr(&sh.pose_pub[lane]);
    int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == kPose + 1), seen = 0, fdone = 0;
    float x, y, yaw;
    if constexpr (GATED) {
      const uint32_t a_go = lds_addr(&sh.gate_open[0]);
      while (lds_peek(a_go) == 0 && --budget > 0) __builtin_amdgcn_s_sleep(1);
      const volatile float *gs = sh.gstate;
      x = gs[0]; y = gs[1]; yaw = gs[2];
    } else {
      x = a.state[0]; y = a.state[1]; yaw = a.state[2];
    }
    for (int t = 0; t < T; t++) {
      while (seen < t + 1 && --budget > 0) {  // rec(t) is written before a dynamics wave publishes step t
        seen = dyn_pub_min<ND>(a_dynw);
        if (seen < t + 1) __builtin_amdgcn_s_sleep(1);
      }
      const float4 r0 = *reinterpret_cast<const float4 *>(&sh.rec[t & (kMRing - 1)][r][0]);  // s3 s4 s5 s6
      float spsi, cpsi;
#ifdef MPPI_DIAG_NOPOSE  // diagnostic build: the hand-overs without the arithmetic (which rider paces the group?)
      spsi = 0.0f; cpsi = 1.0f;
#else
      sincos_fast(yaw, spsi, cpsi);
#endif
      // the ring slot held step t - kMRing.  The fetch wave sets fetch_pub = t' in iteration t', after it has
      // read the record of step t' (none in iteration 0): fetch_pub >= max(t - kMRing, 1) says that record was read
      const int need = (t >= kMRing) ? max(t - kMRing, 1) : 0;
      while (fdone < need && --budget > 0) fdone = lds_peek(a_fp);
      *reinterpret_cast<float4 *>(&sh.pts[t & (kMRing - 1)][r][0]) = make_float4(x, y, cpsi, spsi);
      lds_publish(a_mypub, t + 1);
      // computeKinematics + incrementState for x, y, yaw (neural_net_model.cu:346-355, 334-344)
      const float sd0 = fmaf(cpsi, r0.y, -(spsi * r0.z));
      const float sd1 = fmaf(spsi, r0.y, cpsi * r0.z);
      const float sd2 = a.negate_yaw_der ? -r0.w : r0.w;
      x = fmaf(sd0, a.dt, x);
      y = fmaf(sd1, a.dt, y);
      yaw = fmaf(sd2, a.dt, yaw);
    }
    spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[kPose]));
  } else if (SPLIT && role == kFetch) {
    // -------------------------------- fetch wave (ND = 4): one lane per rollout --------------------------------
    // The two costmap texels of every step.  Software-pipelined by one step: the texels of step t are requested
    // in iteration t and handed to the cost wave in iteration t+1.  (Giving this wave the stabilizing term of
    // computeCost as well -- the cost wave is the heaviest rider left -- was measured: K=16384 87.0 -> 94.5 us.)
    const bool active = lane < NR;
    const int r = active ? lane : NR - 1;
    const uint32_t a_pose = lds_addr(&sh.pose_pub[0]);
    const uint32_t a_cd = lds_addr(&sh.cost_done[0]);
    const uint32_t a_mypub = lds_addr(&sh.fetch_pub[lane]);
    const bool affine = a.cost.affine != 0;
    int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == kFetch + 1), seen = 0, cdone = 0;
    float tf_p = 0.0f, tb_p = 0.0f;
    for (int t = 0; t <= T; t++) {
      float tf = 0.0f, tb = 0.0f;
      if (t < T) {
        while (seen < t + 1 && --budget > 0) {
          seen = lds_peek(a_pose);
          if (seen < t + 1) __builtin_amdgcn_s_sleep(1);
        }
        const float4 p = *reinterpret_cast<const float4 *>(&sh.pts[t & (kMRing - 1)][r][0]);  // x, y, cos psi, sin psi
#ifdef MPPI_DIAG_NOPOSE
        tf = p.x; tb = p.y;
#else
        const float st[3] = {p.x, p.y, 0.0f};
        if (affine) track_fetch<true>(a.cost, st, p.z, p.w, tf, tb);
        else track_fetch<false>(a.cost, st, p.z, p.w, tf, tb);
#endif
      }
      if (t > 0) {
        // the texels of step t-1; their ring slot held step t-1-kMRing, which the cost wave must have consumed
        while (cdone < t - kMRing && --budget > 0) cdone = lds_peek(a_cd);
        *reinterpret_cast<float2 *>(&sh.tex[(t - 1) & (kMRing - 1)][r][0]) = make_float2(tf_p, tb_p);
        lds_publish(a_mypub, t);  // steps < t are out (and the record of step t has been read)
      }
      tf_p = tf; tb_p = tb;
    }
    spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[kFetch]));
  } else if (SPLIT) {
    // -------------------------------- cost wave (ND = 4): one lane per rollout --------------------------------
    const bool active = lane < NR;
    const int r = active ? lane : NR - 1;
    const int k = blockIdx.x * NR + r;
    const uint32_t a_fetch = lds_addr(&sh.fetch_pub[0]);
    const uint32_t a_mydone = lds_addr(&sh.cost_done[lane]);
    const bool ctrl = a.cost.need_control_cost != 0;
    int crash = 0, budget = spin_budget_init(a.spin_budget, T, a.fault_wave == kCost + 1), seen = 0;
    float J = 0.0f;
    for (int t = 0; t < T; t++) {
      const double rt = a.inv_t[t];
      // the fetch wave publishes the texels of step t after the pose wave has read rec(t): the records of step t
      // are there
      while (seen < t + 1 && --budget > 0) {
        seen = lds_peek(a_fetch);
        if (seen < t + 1) __builtin_amdgcn_s_sleep(1);
      }
      const float4 r0 = *reinterpret_cast<const float4 *>(&sh.rec[t & (kMRing - 1)][r][0]);      // s3 s4 s5 s6
      const float4 r1 = *reinterpret_cast<const float4 *>(&sh.ctl_rec[t & (kMRing - 1)][r][0]);  // u0 u1 du0 du1
      const float2 tx = *reinterpret_cast<const float2 *>(&sh.tex[t & (kMRing - 1)][r][0]);      // front, back texel
      lds_publish(a_mydone, t + 1);  // executes after the three reads (the LDS runs a wave's instructions in order)
      const int rc = (int)((t > 0) & (fabsf(r0.x) >= kRollCrash));  // getCrash of update t-1
#ifdef MPPI_DIAG_NOCOST  // diagnostic build, as above
      crash |= rc;
      int crash_new = crash;
      const float Jn = J + r0.y + r1.x + tx.x + (float)rt;
#else
      CostTerms ct;
      if (ctrl) cost_terms_a<true>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
      else cost_terms_a<false>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
      // running mean over 1..T-1 (Q5); the t = 0 evaluation is discarded
      crash |= rc;
      int crash_new = crash;
      const float c = cost_terms_b(a.cost, ct, tx.x, tx.y, crash_new);
      const float Jn = running_mean(J, c, t, rt);
#endif
      J = (t > 0) ? Jn : J;
      crash = (t > 0) ? crash_new : crash;
    }
    // a hand-over that never arrived, in ANY wave of the group: poison, do not hang (mppi_device.hpp)
    {
      // lane r < 8 looks at the finished word of wave r (the cost wave's own counts as set)
      const uint32_t a_fin = lds_addr(&sh.fin[lane & 7]);
      for (;;) {
        const int v = ((lane & 7) == kCost) ? 1 : lds_peek_lanes(a_fin);
        const bool all = __builtin_amdgcn_ballot_w64(v != 0) == ~0ull;
        if (all || --budget <= 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (budget <= 0 || lds_peek(lds_addr(&sh.fail[0])) != 0) J = __builtin_nanf("");
    }
    if (active) a.costs[k] = J + 0.0f;  // + terminalCost (= 0), costs.cu:411-414
    publish_min_cost(a, active ? J + 0.0f : INFINITY);
  } else {
    // -------------------------------- cost wave: one lane per rollout --------------------------------
    // Software-pipelined by one step: the costmap texels of step t are requested in iteration t and
    // consumed in iteration t+1, so their latency never stalls the consumption of the rings.
    const bool active = lane < NR;
    const int r = active ? lane : NR - 1;
    const int k = blockIdx.x * NR + r;
    const uint32_t a_dynw = lds_addr(&sh.dyn_pub[lane & (ND - 1)][0]);
    const uint32_t a_mydone = lds_addr(&sh.cost_done[lane]);
    const bool affine = a.cost.affine != 0, ctrl = a.cost.need_control_cost != 0;
    float x = a.state[0], y = a.state[1], yaw = a.state[2];
    int crash = 0, budget = spin_budget_init(a.spin_budget, T, a.fault_wave == kCost + 1), seen = 0;
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
        // rec(t) is written before a dynamics wave publishes step t; ctl(t) was published before the
        // dynamics waves could start step t
        while (seen < t + 1 && --budget > 0) {
          seen = dyn_pub_min<ND>(a_dynw);
          if (seen < t + 1) __builtin_amdgcn_s_sleep(1);
        }
        const float4 r0 = *reinterpret_cast<const float4 *>(&sh.rec[t & (kMRing - 1)][r][0]);      // s3 s4 s5 s6
        const float4 r1 = *reinterpret_cast<const float4 *>(&sh.ctl_rec[t & (kMRing - 1)][r][0]);  // u0 u1 du0 du1
        lds_publish(a_mydone, t + 1);  // executes after the two reads (the LDS runs a wave's instructions in order)
        rc = (int)((t > 0) & (fabsf(r0.x) >= kRollCrash));  // getCrash of update t-1
        float spsi, cpsi;
        sincos_fast(yaw, spsi, cpsi);
        const float st[3] = {x, y, yaw};
        if (affine) track_fetch<true>(a.cost, st, cpsi, spsi, tf, tb);
        else track_fetch<false>(a.cost, st, cpsi, spsi, tf, tb);
        if (ctrl) cost_terms_a<true>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
        else cost_terms_a<false>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
        // computeKinematics + incrementState for x, y, yaw (neural_net_model.cu:346-355, 334-344)
        const float sd0 = fmaf(cpsi, r0.y, -(spsi * r0.z));
        const float sd1 = fmaf(spsi, r0.y, cpsi * r0.z);
        const float sd2 = a.negate_yaw_der ? -r0.w : r0.w;
        x = fmaf(sd0, a.dt, x);
        y = fmaf(sd1, a.dt, y);
        yaw = fmaf(sd2, a.dt, yaw);
      }
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
    // a hand-over that never arrived, in ANY wave of the group: poison, do not hang (mppi_device.hpp)
    {
      const uint32_t a_fin = lds_addr(&sh.fin[(lane < ND) ? lane : kCtl]);  // lanes 0..ND-1: dynamics, the rest: control
      for (;;) {
        const int v = lds_peek_lanes(a_fin);
        int all = __builtin_amdgcn_readlane(v, ND);
#pragma unroll
        for (int w = 0; w < ND; w++) all &= __builtin_amdgcn_readlane(v, w);
        if (all != 0 || --budget <= 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (budget <= 0 || lds_peek(lds_addr(&sh.fail[0])) != 0) J = __builtin_nanf("");
    }
    if (active) a.costs[k] = J + 0.0f;  // + terminalCost (= 0), costs.cu:411-414
    publish_min_cost(a, active ? J + 0.0f : INFINITY);
  }
}

template <int H, int NHID>
static hipError_t launch_multi_t(const RolloutArgs &a, int nd, hipStream_t stream)
{
  if (a.gate != nullptr) {  // gated: the automatic form only (abi_solve.hip: chain_ok)
    if (nd != 44) return hipErrorInvalidValue;
    MPPI_LAUNCH_ROLLOUT((rollout_multi_kernel<H, NHID, 4, true, true, true>), dim3(a.K / 64), dim3(8 * 64), 0, stream, a);
    return hipGetLastError();
  }
  if (nd == 44) MPPI_LAUNCH_ROLLOUT((rollout_multi_kernel<H, NHID, 4, true, true>), dim3(a.K / 64), dim3(8 * 64), 0, stream, a);  // tree output layer
  else if (nd == 4) MPPI_LAUNCH_ROLLOUT((rollout_multi_kernel<H, NHID, 4>), dim3(a.K / 64), dim3(8 * 64), 0, stream, a);
  else if (nd == 2) MPPI_LAUNCH_ROLLOUT((rollout_multi_kernel<H, NHID, 2>), dim3(a.K / 32), dim3(4 * 64), 0, stream, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// 6-64-64-64-64-4 is left to the other forms: its 284 weight registers per lane do not fit a wave of a
// six-wave workgroup (256 VGPRs) without spilling
bool multi_variant_supported(int hidden, int n_hidden)
{
  return (hidden == 32 && (n_hidden == 2 || n_hidden == 4)) || (hidden == 64 && n_hidden == 2);
}

hipError_t launch_rollout_multi(int hidden, int n_hidden, const RolloutArgs &a, int nd, hipStream_t stream)
{
  if (nd != 2 && nd != 4 && nd != 44) return hipErrorInvalidValue;  // 44: ND = 4 with the tree output layer
  if (a.K % (16 * (nd == 44 ? 4 : nd)) != 0) return hipErrorInvalidValue;
  if (hidden == 32 && n_hidden == 2) return launch_multi_t<32, 2>(a, nd, stream);
  if (hidden == 64 && n_hidden == 2) return launch_multi_t<64, 2>(a, nd, stream);
  if (hidden == 32 && n_hidden == 4) return launch_multi_t<32, 4>(a, nd, stream);
  return hipErrorInvalidValue;
}

}  // namespace mppi

#ifdef MPPI_STAMPS
extern "C" int mppi_debug_read_multi_stamps(unsigned long long *out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mppi::g_multi_stamps), sizeof(unsigned long long) * 8);
}
#endif
