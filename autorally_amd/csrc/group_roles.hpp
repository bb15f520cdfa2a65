// group_roles.hpp -- the wavefronts that ride along with the dynamics wavefronts of a rollout GROUP (16 rollouts
// whose network is split over NW dynamics wavefronts of one workgroup, rollout_oct.hip), written over the
// group's LDS block: pose wave -> cost wave, noise wave -> control wave.  Each is one pipeline stage: a rider
// only gets the issue slots its SIMD's dynamics wave leaves free (the waits of the swaps), and one stage per SIMD
// keeps every stage inside that budget.
//
// SH (the __shared__ struct of the kernel) provides
//   static constexpr int NW, NSW;           dynamics waves of the group; swaps they publish per step
//   static constexpr int kR;                rollouts of the group: 16, or 8 (rollout_row64.hip at K <= 8 x #CUs) -- the riders keep
//                                           their 16-rollout lane layouts; slots j >= kR repeat rollout j - kR (the same values
//                                           to the same LDS addresses) and are masked where a wave touches global memory
//   int   xseq[NW][64];                     swaps published by dynamics wave w (word 0 is read)
//   float rec[kGRing][16][4];               s3..s6 before the update of step t (written by dynamics wave 0
//                                           BEFORE it publishes the first swap of step t)
//   int   cost_done[64];                    steps consumed by the cost wave
//   float ctl_b1[kGRing][64];               layer-0 B operand of k-step 1, [u0c, u1c, 0, 0][g] per rollout
//   float ctl_rec[kGRing][16][4];           clamped u0, u1 and du0, du1 for the cost wave
//   int   ctl_pub[64];                      steps published by the control wave
//   float tex[kGRing][16][2]; int pose_pub[64];   front / back texel of step t; steps published by the pose wave
//   float eps[kGRing][16][2]; int rng_pub[64];    the generator's two uniforms of step t; steps published by the noise wave
//   int   fail[4]; int fin[8];              mppi_device.hpp: spin_finish
// Roles (= wave index in the workgroup): 0..NW-1 dynamics, then pose, cost, control, noise.  a.fault_wave ==
// role + 1 starts that role with an exhausted poll budget (mppi_debug_inject_handover_fault).
//
// Reference: the per-step bookkeeping of rolloutKernel, mppi_controller.cu:127-177 (control perturbation
// :136-153, store before the clamp :155-158 (Q3), running-mean cost :160-166 (Q5)); costs.cu:396-409.
#pragma once

#include "mppi_device.hpp"
#include "noise_device.hpp"

namespace mppi {

constexpr int kGRing = 16;     // steps in flight between the waves of a group (power of two)
constexpr int kGCtlChunk = 4;  // steps of U / explicit eps the control wave requests at once

template <class SH>
struct GroupRoles {
  static constexpr int kPose = SH::NW, kCost = SH::NW + 1, kCtl = SH::NW + 2, kRng = SH::NW + 3;
  static constexpr int kWaves = SH::NW + 4;
  static_assert(kWaves <= (int)(sizeof(SH::fin) / sizeof(int)), "fin[]");
  static_assert(SH::kR == 16 || SH::kR == 8, "rollouts per group");
};
// rollout slot s of a rider's lane layout -> (rollout of the group, first rollout of the launch's numbering, is it real)
template <class SH>
__device__ __forceinline__ int group_j(int slot) { return slot & (SH::kR - 1); }
template <class SH>
__device__ __forceinline__ bool group_real(int slot) { return slot < SH::kR; }

template <class SH>
__device__ __forceinline__ int group_seq_min(SH &sh)
{
  int m = lds_peek(lds_addr(&sh.xseq[0][0]));
#pragma unroll
  for (int w = 1; w < SH::NW; w++) m = min(m, lds_peek(lds_addr(&sh.xseq[w][0])));
  return m;
}

// ---------------------------------- noise wave ----------------------------------
// The handle's MRG32k3a streams: the two uniform draws of every step into a ring for the control wave, which does
// Box-Muller.  The generator's two components are independent third-order recurrences that meet only in the output
// z = (p1 - p2) mod m1, so a rollout takes TWO lanes -- lane 2 j + c runs component c of rollout j with its own multipliers,
// modulus and fold constant in registers -- and the partner's p2 comes by one DPP move: one component's instructions per
// draw instead of two (the 64-bit products are quarter rate, and a rider only gets the issue slots its SIMD's dynamics
// wave leaves).  Integer arithmetic throughout: the same words as mrg_next_z (noise_device.hpp).
struct MrgHalf {
  uint32_t s0, s1, s2;  // component c's three words, oldest first
};
// x < 2^63 -> x mod m, m = 2^32 - C (C = 209: two rounds suffice and the third changes nothing; C = 22853: three)
__device__ __forceinline__ uint32_t fold_m(uint64_t x, uint32_t C, uint32_t m)
{
  x = (x >> 32) * C + (x & 0xffffffffULL);
  x = (x >> 32) * C + (x & 0xffffffffULL);
  x = (x >> 32) * C + (x & 0xffffffffULL);
  if (x >= m) x -= m;
  return (uint32_t)x;
}
// this lane's three generator words (lane 2 j + c: component c of rollout j), requested by a kernel in FRONT of its start
// barrier: they are the head of a launch's critical path (words -> first draws -> Box-Muller -> first controls -> the dynamics
// waves start), and the barrier is a wait for the workgroup's last wave to arrive (row form: first controls 6 070 -> 5 170
// cycles after the first instruction, rollout 47.4 -> 45.6 us)
template <class SH>
__device__ __forceinline__ MrgHalf group_rng_load(const RolloutArgs &a)
{
  const int lane = threadIdx.x & 63;
  const int j = group_j<SH>((lane >> 1) & 15), c = lane & 1;
  const int k = (int)blockIdx.x * SH::kR + j;
  const int K = a.K;
  MrgHalf g{0, 0, 0};
  if (a.inline_noise != 0 && lane < 2 * kRolloutsPerWave) {
    g.s0 = a.rng_in[(3 * c) * K + k]; g.s1 = a.rng_in[(3 * c + 1) * K + k]; g.s2 = a.rng_in[(3 * c + 2) * K + k];
  }
  return g;
}
// PRE: the words were requested by group_rng_load (pre); otherwise they are loaded here
template <class SH, bool PRE = false>
__device__ __forceinline__ void group_rng_wave(const RolloutArgs &a, SH &sh, const MrgHalf pre = MrgHalf{0, 0, 0})
{
  using R = GroupRoles<SH>;
  const int lane = threadIdx.x & 63;
  const int j = group_j<SH>((lane >> 1) & 15), c = lane & 1;
  const int k = (int)blockIdx.x * SH::kR + j;
  const int K = a.K, T = a.T;
  const bool active = lane < 2 * kRolloutsPerWave;
  const bool real = lane < 2 * SH::kR;
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == R::kRng + 1);
  if (a.inline_noise != 0) {
    // p = (A sX - Bn s0) mod m: component 1 = (a12 s11 - a13n s10) mod m1, component 2 = (a21 s22 - a23n s20) mod m2
    const uint32_t m = c ? (uint32_t)kM2 : (uint32_t)kM1, C = c ? kC2 : kC1;
    const uint32_t A = c ? (uint32_t)kA21 : (uint32_t)kA12, Bn = c ? (uint32_t)kA23N : (uint32_t)kA13N;
    MrgHalf g{0, 0, 0};
    if constexpr (PRE) {
      g = pre;
    } else if (active) {
      g.s0 = a.rng_in[(3 * c) * K + k]; g.s1 = a.rng_in[(3 * c + 1) * K + k]; g.s2 = a.rng_in[(3 * c + 2) * K + k];
    }
    const uint32_t a_ctl = lds_addr(&sh.ctl_pub[0]);
    const uint32_t a_mypub = lds_addr(&sh.rng_pub[lane]);
    int seen = 0;
    for (int t = 0; t < T; t++) {
      float2 e = make_float2(0.5f, 0.5f);
#ifdef MPPI_DIAG_NORNG  // diagnostic build: the hand-overs without the generator steps (which rider paces the group?)
      e = make_float2(0.25f + 0.001f * (float)t, 0.5f);
#else
#pragma unroll
      for (int d = 0; d < 2; d++) {  // one timestep's pair of uniforms: two generator steps (uniform_pair)
        const uint32_t p = fold_m((uint64_t)A * (c ? g.s2 : g.s1) + (uint64_t)Bn * (m - g.s0), C, m);
        g.s0 = g.s1; g.s1 = g.s2; g.s2 = p;
        const uint32_t p2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)p, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]: lane ^ 1
        uint32_t z = (p >= p2) ? p - p2 : p + (uint32_t)kM1 - p2;  // (even lanes: p = p1)
        if (z == 0) z = (uint32_t)kM1;
        const float u = (float)z * 0x1p-32f;
        if (d == 0) e.x = u; else e.y = u;
      }
#endif
      // slot t % kGRing held step t - kGRing, consumed once the control wave has published that step
      const int need = t - kGRing + 1;
      while (seen < need && --budget > 0) {
        seen = lds_peek(a_ctl);
        if (seen < need) __builtin_amdgcn_s_sleep(2);
      }
      if (active && c == 0) *reinterpret_cast<float2 *>(&sh.eps[t & (kGRing - 1)][j][0]) = e;
      lds_publish(a_mypub, t + 1);
    }
    if (real) {
      a.rng_out[(3 * c) * K + k] = g.s0; a.rng_out[(3 * c + 1) * K + k] = g.s1; a.rng_out[(3 * c + 2) * K + k] = g.s2;
    }
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[R::kRng]));
}

// ---------------------------------- control wave ----------------------------------
// FOUR steps per iteration, one lane per (step, rollout): lane = 16 q + j works on step t0 + q of rollout j.  Nothing
// of mppi_controller.cu:136-153 depends on the state or on the previous step, and a rider only gets the issue slots
// its SIMD's dynamics wave leaves free: with one step per iteration on 16 lanes the Box-Muller chain (a division, a
// square root, three polynomials: ~100 dependent instructions) made this wave the pace of the whole group (row form,
// K=4096, T=100: 68.7 us; 55.6 us with the transform removed).  Per step it now issues a quarter of that.
// a_gate_open != 0 (gated launch, rollout_row.hip): the nominal sequence is in the gate block, host-written: the wave waits for
// the group's pose wave to have seen the gate open (that LDS word) and reads U with system-scope loads
template <class SH>
__device__ __forceinline__ void group_control_wave(const RolloutArgs &a, SH &sh, const uint32_t a_gate_open = 0)
{
  using R = GroupRoles<SH>;
  constexpr int NSW = SH::NSW;
  static_assert(kGCtlChunk == 4 && kGRing >= 2 * kGCtlChunk, "four steps per iteration, lanes 16 q + j");
  const int lane = threadIdx.x & 63;
  const int j = group_j<SH>(lane & 15), q = lane >> 4;
  const bool real = group_real<SH>(lane & 15);
  const int k = (int)blockIdx.x * SH::kR + j;
  const int K = a.K, T = a.T;
  const bool inl = a.inline_noise != 0;
  float2 *const noise = reinterpret_cast<float2 *>(a.noise);
  const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
  const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
  const bool pure_noise_k = (k >= a.k99);  // :141
  const uint32_t a_cd = lds_addr(&sh.cost_done[0]);
  const uint32_t a_mypub = lds_addr(&sh.ctl_pub[lane]);
  const uint32_t a_rng = lds_addr(&sh.rng_pub[0]);
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == R::kCtl + 1);
  const bool gated = a_gate_open != 0;
  if (gated)
    while (lds_peek(a_gate_open) == 0 && --budget > 0) __builtin_amdgcn_s_sleep(1);
  int seen_x = 0, seen_c = 0, seen_r = 0;  // swaps published by all dynamics waves / steps consumed by the cost wave / pairs drawn
  // The first iteration is ONE step (lanes of q = 0 only), so that the dynamics waves can start as soon as the noise
  // wave has drawn its first pair instead of its first four (~1 us of every launch); four steps from then on.
  for (int t0 = 0, n = 1; t0 < T; t0 += n, n = kGCtlChunk) {
    const int t = t0 + q;
    const bool live = (t < T) & (q < n);
    const int tl = live ? t : T - 1;
    // this lane's nominal control and (explicit noise) eps
    float2 Ut;
    if (gated) {
      const unsigned long long ub = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(Useq + tl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      Ut = make_float2(__uint_as_float((unsigned)ub), __uint_as_float((unsigned)(ub >> 32)));
    } else {
      Ut = Useq[tl];
    }
    float2 e = (live && !inl) ? noise[(size_t)tl * K + k] : make_float2(0.0f, 0.0f);
    // the last slot of the chunk: slot t % kGRing held step t - kGRing -- the dynamics waves read it during step
    // t - kGRing - 1 (done once all of them published the first swap of step t - kGRing), the cost wave in step t - kGRing
    const int tm = min(t0 + n, T) - 1;
    const int need_x = (tm >= kGRing) ? (tm - kGRing) * NSW + 1 : 0;
    const int need_c = tm - kGRing + 1;
    while ((seen_x < need_x || seen_c < need_c) && --budget > 0) {
      seen_x = group_seq_min(sh);
      seen_c = lds_peek(a_cd);
      if (seen_x < need_x || seen_c < need_c) __builtin_amdgcn_s_sleep(2);
    }
    if (inl) {  // generator mode: the pairs of the chunk's steps from the noise wave's ring
      while (seen_r < tm + 1 && --budget > 0) {
        seen_r = lds_peek(a_rng);
        if (seen_r < tm + 1) __builtin_amdgcn_s_sleep(1);
      }
#ifdef MPPI_DIAG_NOBM  // diagnostic build: no Box-Muller
      e = *reinterpret_cast<const float2 *>(&sh.eps[tl & (kGRing - 1)][j][0]);
#else
      e = box_muller(*reinterpret_cast<const float2 *>(&sh.eps[tl & (kGRing - 1)][j][0]));
#endif
    }
    if (live) {
      // control perturbation, mppi_controller.cu:136-153
      const bool nf = noise_free_k | (t < a.opt_delay);
      const float n0 = e.x * a.nu[0], n1 = e.y * a.nu[1];
      const float du0 = nf ? 0.0f : n0, du1 = nf ? 0.0f : n1;
      float u0 = nf ? Ut.x : (pure_noise_k ? n0 : Ut.x + n0);
      float u1 = nf ? Ut.y : (pure_noise_k ? n1 : Ut.y + n1);
      if (real) noise[(size_t)t * K + k] = make_float2(u0, u1);  // before the clamp (Q3)
      u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
      u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
      const int slot = t & (kGRing - 1);
      sh.ctl_b1[slot][j] = u0;
      sh.ctl_b1[slot][kRolloutsPerWave + j] = u1;
      *reinterpret_cast<float4 *>(&sh.ctl_rec[slot][j][0]) = make_float4(u0, u1, du0, du1);
    }
    // after the reads of eps(t0 .. tm) and the records of all four steps (the LDS runs a wave's instructions in order):
    // releases those eps slots to the noise wave
    lds_publish(a_mypub, tm + 1);
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[R::kCtl]));
}

// ---------------------------------- pose wave ----------------------------------
// x, y, yaw of the group's rollouts and the two costmap texels per step.  Software-pipelined by one step: the
// texels of step t are requested in iteration t and handed to the cost wave in iteration t+1.
template <class SH, bool AFFINE>
__device__ __forceinline__ void group_pose_wave(const RolloutArgs &a, SH &sh)
{
  using R = GroupRoles<SH>;
  constexpr int NSW = SH::NSW;
  static_assert(SH::kR == 16, "one step per iteration: 16-rollout groups only");
  const int lane = threadIdx.x & 63;
  const int j = lane & 15;
  const int T = a.T;
  const uint32_t a_seq0 = lds_addr(&sh.xseq[0][0]);
  const uint32_t a_cd = lds_addr(&sh.cost_done[0]);
  const uint32_t a_mypub = lds_addr(&sh.pose_pub[lane]);
  float x = a.state[0], y = a.state[1], yaw = a.state[2];
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == R::kPose + 1), seen = 0, cdone = 0;
  float tf_p = 0.0f, tb_p = 0.0f;
  for (int t = 0; t <= T; t++) {
    float tf = 0.0f, tb = 0.0f;
    if (t < T) {
      // rec(t) is written before wave 0 publishes the first swap of step t (forms whose dynamics waves each write
      // the records of their own rollouts, SH::kRecByAll: before every one of them has published step t)
      const int need = t * NSW + 1;
      while (seen < need && --budget > 0) {
        seen = SH::kRecByAll ? group_seq_min(sh) : lds_peek(a_seq0);
        if (seen < need) __builtin_amdgcn_s_sleep(1);
      }
      const float4 r0 = *reinterpret_cast<const float4 *>(&sh.rec[t & (kGRing - 1)][j][0]);  // s3 s4 s5 s6
      float spsi, cpsi;
#ifdef MPPI_DIAG_NOPOSE  // diagnostic build: no sin/cos, no texel fetches
      spsi = 0.0f; cpsi = 1.0f; tf = x; tb = y;
#else
      sincos_fast(yaw, spsi, cpsi);
      const float st[3] = {x, y, yaw};
      track_fetch<AFFINE>(a.cost, st, cpsi, spsi, tf, tb);
#endif
      // computeKinematics + incrementState for x, y, yaw (neural_net_model.cu:346-355, 334-344)
      const float sd0 = fmaf(cpsi, r0.y, -(spsi * r0.z));
      const float sd1 = fmaf(spsi, r0.y, cpsi * r0.z);
      const float sd2 = a.negate_yaw_der ? -r0.w : r0.w;
      x = fmaf(sd0, a.dt, x);
      y = fmaf(sd1, a.dt, y);
      yaw = fmaf(sd2, a.dt, yaw);
    }
    if (t > 0) {
      // the texels of step t-1; their ring slot held step t-1-kGRing, which the cost wave must have consumed
      while (cdone < t - kGRing && --budget > 0) cdone = lds_peek(a_cd);
      if (lane < kRolloutsPerWave) *reinterpret_cast<float2 *>(&sh.tex[(t - 1) & (kGRing - 1)][lane][0]) = make_float2(tf_p, tb_p);
      lds_publish(a_mypub, t);  // steps < t are out
    }
    tf_p = tf; tb_p = tb;
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[R::kPose]));
}

// ---------------------------------- pose wave, four steps per iteration ----------------------------------
// lane = 4 j + q works on step t0 + q of rollout j.  The sin/cos pair and the two texel addresses of a step -- nine tenths
// of the pose wave's instructions -- depend only on that step's pose, so four steps of a rollout run side by side; what is
// sequential, the three Euler chains x' = fma(sd0, dt, x), y' = .., yaw' = .. (incrementState, neural_net_model.cu:334-344),
// runs link by link with the link's derivative broadcast inside the quad (one DPP move per link): the same operations in
// the same order as one step per iteration, hence the same bits.  Why: with the activations of the row form travelling by
// DPP (rollout_row.hip) the dynamics waves no longer leave the issue slots a one-step-per-iteration pose wave needs --
// it had become the pace of the group (rollout 52.0 us, 49.5 us with its arithmetic removed).
template <int Q>
__device__ __forceinline__ float quad_bc(float v)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xF, 0xF, false));  // quad_perm [Q, Q, Q, Q]
}
// the value before each of the four links of v' = fma(d, dt, v) (this lane's: vq) and after the last (v_out)
__device__ __forceinline__ void quad_chain(float v, float d, float dt, int q, float &vq, float &v_out)
{
  const float v1 = fmaf(quad_bc<0>(d), dt, v);
  const float v2 = fmaf(quad_bc<1>(d), dt, v1);
  const float v3 = fmaf(quad_bc<2>(d), dt, v2);
  v_out = fmaf(quad_bc<3>(d), dt, v3);
  const float lo = (q == 0) ? v : v1, hi = (q == 2) ? v2 : v3;
  vq = (q < 2) ? lo : hi;
}
template <class SH, bool AFFINE>
__device__ __forceinline__ void group_pose_wave4(const RolloutArgs &a, SH &sh, const float x0, const float y0, const float yaw0,
                                                 const int budget_cut = 0);
template <class SH, bool AFFINE>
__device__ __forceinline__ void group_pose_wave4(const RolloutArgs &a, SH &sh)
{
  group_pose_wave4<SH, AFFINE>(a, sh, a.state[0], a.state[1], a.state[2]);
}
// x0, y0, yaw0: the pose the rollouts start from (a.state[0..2], or what a gated kernel received through its gate block);
// budget_cut != 0: a wait in front of this call already ran out -- the wave starts with an exhausted poll budget
template <class SH, bool AFFINE>
__device__ __forceinline__ void group_pose_wave4(const RolloutArgs &a, SH &sh, const float x0, const float y0, const float yaw0,
                                                 const int budget_cut)
{
  using R = GroupRoles<SH>;
  constexpr int NSW = SH::NSW;
  static_assert(kRolloutsPerWave == 16 && kGRing >= 8, "lane = 4 j + q");
  const int lane = threadIdx.x & 63;
  const int j = group_j<SH>(lane >> 2), q = lane & 3;
  const int T = a.T;
  const uint32_t a_seq0 = lds_addr(&sh.xseq[0][0]);
  const uint32_t a_cd = lds_addr(&sh.cost_done[0]);
  const uint32_t a_mypub = lds_addr(&sh.pose_pub[lane]);
  float x = x0, y = y0, yaw = yaw0;  // the pose before step t0, the same in the four lanes of a rollout
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == R::kPose + 1 || budget_cut != 0), seen = 0, cdone = 0;
  for (int t0 = 0; t0 < T; t0 += 4) {
    const int t = t0 + q;
    const int tend = min(t0 + 4, T);  // the chunk is steps [t0, tend); lanes of later steps compute on whatever the ring holds
    // rec(t) is written before wave 0 publishes the first swap of step t (forms whose dynamics waves each write
    // the records of their own rollouts, SH::kRecByAll: before every one of them has published step t)
    const int need = (tend - 1) * NSW + 1;
    while (seen < need && --budget > 0) {
      seen = SH::kRecByAll ? group_seq_min(sh) : lds_peek(a_seq0);
      if (seen < need) __builtin_amdgcn_s_sleep(1);
    }
    const float4 r0 = *reinterpret_cast<const float4 *>(&sh.rec[t & (kGRing - 1)][j][0]);  // s3 s4 s5 s6
    // computeKinematics + incrementState for x, y, yaw (neural_net_model.cu:346-355, 334-344)
    float yaw_q, yaw_n, x_q, x_n, y_q, y_n;
    quad_chain(yaw, a.negate_yaw_der ? -r0.w : r0.w, a.dt, q, yaw_q, yaw_n);
    float spsi, cpsi, tf = 0.0f, tb = 0.0f;
#ifdef MPPI_DIAG_NOPOSE  // diagnostic build: no sin/cos, no texel fetches
    spsi = 0.0f; cpsi = 1.0f;
#else
    sincos_fast(yaw_q, spsi, cpsi);
#endif
    quad_chain(x, fmaf(cpsi, r0.y, -(spsi * r0.z)), a.dt, q, x_q, x_n);
    quad_chain(y, fmaf(spsi, r0.y, cpsi * r0.z), a.dt, q, y_q, y_n);
#ifdef MPPI_DIAG_NOPOSE
    tf = x_q; tb = y_q;
#else
    const float st[3] = {x_q, y_q, yaw_q};
    track_fetch<AFFINE>(a.cost, st, cpsi, spsi, tf, tb);
#endif
    x = x_n; y = y_n; yaw = yaw_n;
    // the texels' ring slots held steps t - kGRing, which the cost wave must have consumed
    while (cdone < tend - kGRing && --budget > 0) cdone = lds_peek(a_cd);
    if (t < T) *reinterpret_cast<float2 *>(&sh.tex[t & (kGRing - 1)][j][0]) = make_float2(tf, tb);
    lds_publish(a_mypub, tend);  // steps < tend are out
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[R::kPose]));
}

// the end of the cost wave: wait for every other wave's finished word, poison on a raised fail word
template <class SH>
__device__ __forceinline__ float group_settle(SH &sh, int budget, float J)
{
  using R = GroupRoles<SH>;
  const int lane = threadIdx.x & 63;
  // a hand-over that never arrived, in ANY wave of the group: poison, do not hang (mppi_device.hpp).  The
  // other waves raise the fail word before their finished word; the kernel cannot end before they do.
  // lane r < kWaves looks at finished word r (the cost wave's own counts as set)
  const uint32_t a_fin = lds_addr(&sh.fin[(lane < R::kWaves) ? lane : 0]);
  for (;;) {
    const int v = (lane == R::kCost) ? 1 : lds_peek_lanes(a_fin);
    const bool all = __builtin_amdgcn_ballot_w64(v != 0) == ~0ull;
    if (all || --budget <= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  if (budget <= 0 || lds_peek(lds_addr(&sh.fail[0])) != 0) J = __builtin_nanf("");
  return J;
}

// ---------------------------------- cost wave ----------------------------------
// computeCost over the records of the dynamics, control and pose waves; running mean; writes costs[k].
template <class SH, bool CTRL>
__device__ __forceinline__ void group_cost_wave(const RolloutArgs &a, SH &sh)
{
  using R = GroupRoles<SH>;
  const int lane = threadIdx.x & 63;
  static_assert(SH::kR == 16, "one step per iteration: 16-rollout groups only");
  const int j = lane & 15;
  const int k = (int)blockIdx.x * kRolloutsPerWave + j;
  const int T = a.T;
  const uint32_t a_mydone = lds_addr(&sh.cost_done[lane]);
  const uint32_t a_pose = lds_addr(&sh.pose_pub[0]);
  int crash = 0, budget = spin_budget_init(a.spin_budget, T, a.fault_wave == R::kCost + 1), seen = 0;
  float J = 0.0f;
  for (int t = 0; t < T; t++) {
    const double rt = a.inv_t[t];
    // the pose wave publishes the texels of step t after it has read rec(t): the records of step t are there
    // (ctl(t) was published before the dynamics waves could start step t)
    while (seen < t + 1 && --budget > 0) {
      seen = lds_peek(a_pose);
      if (seen < t + 1) __builtin_amdgcn_s_sleep(1);
    }
    const float4 r0 = *reinterpret_cast<const float4 *>(&sh.rec[t & (kGRing - 1)][j][0]);      // s3 s4 s5 s6
    const float4 r1 = *reinterpret_cast<const float4 *>(&sh.ctl_rec[t & (kGRing - 1)][j][0]);  // u0 u1 du0 du1
    const float2 tx = *reinterpret_cast<const float2 *>(&sh.tex[t & (kGRing - 1)][j][0]);      // front, back texel
    lds_publish(a_mydone, t + 1);  // executes after the three reads (the LDS runs a wave's instructions in order)
    const int rc = (int)((t > 0) & (fabsf(r0.x) >= kRollCrash));  // getCrash of update t-1
#ifdef MPPI_DIAG_NOCOST  // diagnostic build: no cost arithmetic
    crash |= rc;
    int crash_new = crash;
    const float Jn = J + r0.y + r1.x + tx.x + (float)rt;
#else
    CostTerms ct;
    cost_terms_a<CTRL>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
    // running mean over 1..T-1 (Q5); the t = 0 evaluation is discarded
    crash |= rc;
    int crash_new = crash;
    const float c = cost_terms_b(a.cost, ct, tx.x, tx.y, crash_new);
    const float Jn = running_mean(J, c, t, rt);
#endif
    J = (t > 0) ? Jn : J;
    crash = (t > 0) ? crash_new : crash;
  }
  J = group_settle(sh, budget, J);
  a.costs[k] = J + 0.0f;  // + terminalCost (= 0), costs.cu:411-414
}

// ---------------------------------- cost wave, four steps per iteration ----------------------------------
// lane = 4 j + q works on step t0 + q of rollout j (the pose wave's layout).  The cost terms of a step depend on that
// step's records only; what is sequential -- the sticky crash flag (an inclusive OR over the steps) and the running mean
// (mppi_controller.cu:160-166) -- runs link by link over the quad, the link's operand broadcast by one DPP move: the
// reference's operations in the reference's order, hence the bits of the one-step-per-iteration wave.
template <int Q>
__device__ __forceinline__ int quad_bc_i(int v)
{
  return __builtin_amdgcn_mov_dpp(v, Q * 0x55, 0xF, 0xF, false);
}
template <class SH, bool CTRL>
__device__ __forceinline__ void group_cost_wave4(const RolloutArgs &a, SH &sh)
{
  using R = GroupRoles<SH>;
  const int lane = threadIdx.x & 63;
  const int j = group_j<SH>(lane >> 2), q = lane & 3;
  const int k = (int)blockIdx.x * SH::kR + j;
  const int T = a.T;
  const uint32_t a_mydone = lds_addr(&sh.cost_done[lane]);
  const uint32_t a_pose = lds_addr(&sh.pose_pub[0]);
  int crash = 0, budget = spin_budget_init(a.spin_budget, T, a.fault_wave == R::kCost + 1), seen = 0;
  float J = 0.0f;  // crash, J: the values before step t0, the same in the four lanes of a rollout
  for (int t0 = 0; t0 < T; t0 += 4) {
    const int t = t0 + q;
    const int tend = min(t0 + 4, T);
    // the pose wave publishes the texels of a step after it has read that step's state record: the records of the chunk
    // are there (ctl(t) was published before the dynamics waves could start step t)
    while (seen < tend && --budget > 0) {
      seen = lds_peek(a_pose);
      if (seen < tend) __builtin_amdgcn_s_sleep(1);
    }
    const float4 r0 = *reinterpret_cast<const float4 *>(&sh.rec[t & (kGRing - 1)][j][0]);      // s3 s4 s5 s6
    const float4 r1 = *reinterpret_cast<const float4 *>(&sh.ctl_rec[t & (kGRing - 1)][j][0]);  // u0 u1 du0 du1
    const float2 tx = *reinterpret_cast<const float2 *>(&sh.tex[t & (kGRing - 1)][j][0]);      // front, back texel
    lds_publish(a_mydone, tend);  // executes after the three reads (the LDS runs a wave's instructions in order)
    const bool counted = (t > 0) & (t < T);  // the t = 0 evaluation is discarded (Q5); lanes beyond T hold stale records
    // getCrash of update t-1 (costs.cu:301-305), then the track term's flag (:402 evaluates the track first)
    int e = (int)(fabsf(r0.x) >= kRollCrash) | (int)(tx.x >= a.cost.boundary_threshold) | (int)(tx.y >= a.cost.boundary_threshold);
    e = counted ? e : 0;
    const int c0 = crash | quad_bc_i<0>(e), c1 = c0 | quad_bc_i<1>(e), c2 = c1 | quad_bc_i<2>(e), c3 = c2 | quad_bc_i<3>(e);
    const int lo = (q == 0) ? c0 : c1, hi = (q == 2) ? c2 : c3;
    int crash_q = (q < 2) ? lo : hi;  // the flag cost_terms_b sees at step t
    crash = c3;
#ifdef MPPI_DIAG_NOCOST  // diagnostic build: no cost arithmetic
    const float c = r0.y + r1.x + tx.x + (float)crash_q;
#else
    CostTerms ct;
    cost_terms_a<CTRL>(a.cost, a.nu, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, ct);
    const float c = cost_terms_b(a.cost, ct, tx.x, tx.y, crash_q);  // ORs the track flag in again: no change
#endif
    // running mean over 1..T-1 (Q5), one link per step of the chunk
#pragma unroll
    for (int l = 0; l < 4; l++) {
      const int tl = t0 + l;  // wave-uniform
      const float cl = (l == 0) ? quad_bc<0>(c) : (l == 1) ? quad_bc<1>(c) : (l == 2) ? quad_bc<2>(c) : quad_bc<3>(c);
      if (tl > 0 && tl < T) {
#ifdef MPPI_DIAG_NOCOST
        J = J + cl + (float)a.inv_t[tl];
#else
        J = running_mean(J, cl, tl, a.inv_t[tl]);
#endif
      }
    }
  }
  J = group_settle(sh, budget, J);
  if (q == 0 && group_real<SH>(lane >> 2)) a.costs[k] = J + 0.0f;  // + terminalCost (= 0), costs.cu:411-414
}

}  // namespace mppi
