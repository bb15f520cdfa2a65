// rollout_bf.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) with the reference's second
// dynamics family, GeneralizedLinear<CarBasisFuncs,7,2,25,CarKinematics,3>
// (PI/generalized_linear.cu:169-245, PI/car_bfs.cuh:44-120), SURVEY 8f row f3.
//
// One lane per rollout: a step is 25 scalar basis functions (one sincos, two f64 divides and a few
// fp32 divides shared by all of them) and a 4 x 25 matrix-vector product -- 100 MACs, nothing a matrix
// instruction could help with -- followed by the same cost evaluation as the network kernels.  W
// (400 B) is staged into LDS and read as broadcasts.  Noise comes from the stand-alone generator.
#include "basis_funcs.hpp"
#include "noise_device.hpp"
#include "mppi_kernels.hpp"

namespace mppi {

constexpr int kBfLanes = 64;

// Device form of the shared sub-expressions.  tan(atan(q) - u0) is evaluated through
// tan(a - b) = (tan a - tan b) / (1 + tan a tan b) with tan(atan q) = q, and sin u0 / tan u0 come from
// one sincos_fast: no atanf, no tanf, no large-argument reduction on the recurrence.  The reference's
// own device code composes CUDA's sinf / atanf / tanf (2-4 ulp each); this form stays within the same
// few ulp of the exact value (tests/test_basis_funcs.py: <= 2e-5 of the derivative's scale against the
// literal restatement).
__device__ __forceinline__ void basis_shared_fast(const float *s, float u0, BasisShared &c)
{
  float q, sn, cs;
  basis_shared_common(s, c, q);
  sincos_fast(u0, sn, cs);
  const float t = sn / cs;
  c.su = sn;
  c.A = c.big ? (q - t) / fmaf(q, t, 1.0f) : -t;
}

// W phi on the device: the same four i-mod-4 chains per output as basis_dynamics (basis_funcs.hpp), two
// outputs per packed multiply-add.  Wr is W transposed, [25] columns of four outputs, held in registers
// for the whole rollout (100 VGPRs; one wavefront per SIMD has 512), loaded once through LDS.
struct BfWeights {
  f32x4 col[kNumBfs];
  __device__ __forceinline__ void load(const float *Wt_s)
  {
#pragma unroll
    for (int i = 0; i < kNumBfs; i++) col[i] = *reinterpret_cast<const f32x4 *>(Wt_s + 4 * i);
  }
};
__device__ __forceinline__ void basis_dynamics_dev(const BfWeights &Wr, const float *phi, float *d)
{
  f32x2 acc01[kBfYThreads], acc23[kBfYThreads];
#pragma unroll
  for (int y = 0; y < kBfYThreads; y++) acc01[y] = acc23[y] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < kNumBfs; i++) {
    const f32x4 w = Wr.col[i];
    const f32x2 p = {phi[i], phi[i]};
    acc01[i % kBfYThreads] = __builtin_elementwise_fma(f32x2{w[0], w[1]}, p, acc01[i % kBfYThreads]);
    acc23[i % kBfYThreads] = __builtin_elementwise_fma(f32x2{w[2], w[3]}, p, acc23[i % kBfYThreads]);
  }
  f32x2 s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
  for (int y = 0; y < kBfYThreads; y++) {
    s01 = s01 + acc01[y];
    s23 = s23 + acc23[y];
  }
  d[0] = s01.x; d[1] = s01.y; d[2] = s23.x; d[3] = s23.y;
}

// computeStateDeriv: kinematics with the yaw rate always negated (generalized_linear.cu:212-217)
__device__ __forceinline__ void bf_state_deriv(const BfWeights &W_s, const float *s, float u0, float u1, float cpsi,
                                               float spsi, float *sd)
{
  sd[0] = fmaf(cpsi, s[4], -(spsi * s[5]));
  sd[1] = fmaf(spsi, s[4], cpsi * s[5]);
  sd[2] = -s[6];
  float phi[kNumBfs];
  BasisShared c;
  basis_shared_fast(s, u0, c);
  basis_funcs_from(s, u1, c, phi);
  basis_dynamics_dev(W_s, phi, sd + 3);
}

__global__ __launch_bounds__(kBfLanes) void rollout_bf_kernel(const RolloutArgs a)
{
  __shared__ __attribute__((aligned(16))) float W_s[4 * kNumBfs];  // transposed: [25][4]
  const int lane = threadIdx.x;
  for (int i = lane; i < 4 * kNumBfs; i += kBfLanes) W_s[(i % kNumBfs) * 4 + i / kNumBfs] = a.wpack[i];
  __syncthreads();
  const int k = blockIdx.x * kBfLanes + lane;
  if (k >= a.K) return;  // K % 64 == 0: never splits a wave
  BfWeights Wr;
  Wr.load(W_s);

  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = a.state[i];
  int crash = 0;
  float J = 0.0f;
  const int K = a.K, T = a.T;
  float2 *const noise = reinterpret_cast<float2 *>(a.noise);
  const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
  const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
  const bool pure_noise_k = (k >= a.k99);  // :141

  float2 e_next = noise[(size_t)k];
  for (int t = 0; t < T; t++) {
    const float2 e = e_next;
    if (t + 1 < T) e_next = noise[(size_t)(t + 1) * K + k];
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
    noise[(size_t)t * K + k] = make_float2(u0, u1);  // before the clamp (Q3)
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
    bf_state_deriv(Wr, s, u0, u1, cpsi, spsi, sd);
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

// ---------------------------------------------------------------------------------------------
// rollout_bf2_kernel: the same rollout with the work of 64 rollouts split over TWO wavefronts of one
// workgroup, as in the quad form of the network kernel (rollout_mfma.hip): the T-step recurrence of a
// rollout is latency bound (K = 2560 gives 40 wavefronts for 1024 SIMDs), so everything that does not
// feed the next state leaves the wavefront that computes it.
//   wave 0 "dynamics": controls + clamp + the 25 basis functions + W phi + Euler update of
//          [roll, u_x, u_y, yaw_mder] (the basis functions do not read x, y, yaw);
//   wave 1 "cost":     x, y, yaw kinematics, sin/cos, the costmap fetches, MPPICosts::computeCost, the
//          running mean and the crash flags, software-pipelined by one step around the fetches, fed by
//          a ring of per-step records (s3..s6 before the update, clamped u, du).
// One-directional hand-over through LDS sequence words (mppi_device.hpp), kBfRing steps deep.
// Arithmetic and its order are those of rollout_bf_kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr int kBfRing = 16;  // power of two

__global__ __launch_bounds__(2 * kBfLanes) void rollout_bf2_kernel(const RolloutArgs a)
{
  __shared__ __attribute__((aligned(16))) float W_s[4 * kNumBfs];  // transposed: [25][4]
  __shared__ float rec[kBfRing][8][kBfLanes];  // [slot][field][lane]: s3 s4 s5 s6 u0 u1 du0 du1
  __shared__ int pub[kBfLanes], done[kBfLanes], fail[4], fin[4];
  const int lane = threadIdx.x & 63;
  const int role = threadIdx.x >> 6;  // wave-uniform
  for (int i = threadIdx.x; i < 4 * kNumBfs; i += 2 * kBfLanes) W_s[(i % kNumBfs) * 4 + i / kNumBfs] = a.wpack[i];
  if (role == 0) { pub[lane] = 0; done[lane] = 0; fail[lane & 3] = 0; fin[lane & 3] = 0; }
  __syncthreads();  // the only barrier
  const int k = blockIdx.x * kBfLanes + lane;
  const int K = a.K, T = a.T;
  const uint32_t a_pub = lds_addr(&pub[0]), a_done = lds_addr(&done[0]);
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == role + 1);  // mppi_device.hpp

  if (role == 0) {
    // ------------------------------ dynamics wave ------------------------------
    const uint32_t a_mypub = lds_addr(&pub[lane]);
    BfWeights Wr;
    Wr.load(W_s);
    float s[kStateDim];
#pragma unroll
    for (int i = 0; i < kStateDim; i++) s[i] = a.state[i];
    float2 *const noise = reinterpret_cast<float2 *>(a.noise);
    const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
    const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
    const bool pure_noise_k = (k >= a.k99);  // :141
    // eps and U are requested two steps ahead: a step of this wave is shorter than an L2 round trip
    float2 e_n1 = noise[(size_t)k], e_n2 = noise[(size_t)min(1, T - 1) * K + k];
    float2 U_n1 = Useq[0], U_n2 = Useq[min(1, T - 1)];
    int seen = 0;  // steps the cost wave has consumed
    for (int t = 0; t < T; t++) {
      const float2 e = e_n1, Ut = U_n1;
      const int tn = min(t + 2, T - 1);
      e_n1 = e_n2;
      U_n1 = U_n2;
      e_n2 = noise[(size_t)tn * K + k];
      U_n2 = Useq[tn];
      float du0, du1, u0, u1;
      if (noise_free_k || t < a.opt_delay) {
        du0 = 0.0f; du1 = 0.0f; u0 = Ut.x; u1 = Ut.y;
      } else {
        du0 = e.x * a.nu[0];
        du1 = e.y * a.nu[1];
        u0 = pure_noise_k ? du0 : Ut.x + du0;
        u1 = pure_noise_k ? du1 : Ut.y + du1;
      }
      noise[(size_t)t * K + k] = make_float2(u0, u1);  // before the clamp (Q3)
      u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
      u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
      while (seen < t - kBfRing + 1 && --budget > 0) seen = lds_peek(a_done);
      const int slot = t & (kBfRing - 1);
      rec[slot][0][lane] = s[3]; rec[slot][1][lane] = s[4]; rec[slot][2][lane] = s[5]; rec[slot][3][lane] = s[6];
      rec[slot][4][lane] = u0;   rec[slot][5][lane] = u1;   rec[slot][6][lane] = du0;  rec[slot][7][lane] = du1;
      lds_publish(a_mypub, t + 1);
      float phi[kNumBfs], d[4];
      BasisShared c;
      basis_shared_fast(s, u0, c);
      basis_funcs_from(s, u1, c, phi);
      basis_dynamics_dev(Wr, phi, d);
#pragma unroll
      for (int i = 0; i < 4; i++) s[3 + i] = fmaf(d[i], a.dt, s[3 + i]);
    }
    spin_finish(budget, lds_addr(&fail[0]), lds_addr(&fin[0]));
  } else {
    // -------------------------------- cost wave --------------------------------
    const uint32_t a_mydone = lds_addr(&done[lane]);
    float x = a.state[0], y = a.state[1], yaw = a.state[2];
    int crash = 0, seen = 0;
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
        while (seen < t + 1 && --budget > 0) {
          seen = lds_peek(a_pub);
          if (seen < t + 1) __builtin_amdgcn_s_sleep(1);
        }
        const int slot = t & (kBfRing - 1);
        const float r3 = rec[slot][0][lane], r4 = rec[slot][1][lane], r5 = rec[slot][2][lane], r6 = rec[slot][3][lane];
        const float u0 = rec[slot][4][lane], u1 = rec[slot][5][lane], du0 = rec[slot][6][lane], du1 = rec[slot][7][lane];
        lds_publish(a_mydone, t + 1);  // executes after the reads (the LDS runs a wave's instructions in order)
        rc = (int)((t > 0) & (fabsf(r3) >= kRollCrash));  // getCrash of update t-1
        float spsi, cpsi;
        sincos_fast(yaw, spsi, cpsi);
        const float st[3] = {x, y, yaw};
        if (t > 0) {
          if (a.cost.affine) track_fetch<true>(a.cost, st, cpsi, spsi, tf, tb);
          else track_fetch<false>(a.cost, st, cpsi, spsi, tf, tb);
        }
        if (a.cost.need_control_cost) cost_terms_a<true>(a.cost, a.nu, r4, r5, u0, u1, du0, du1, ct);
        else cost_terms_a<false>(a.cost, a.nu, r4, r5, u0, u1, du0, du1, ct);
        // computeKinematics (generalized_linear.cu:212-217, yaw rate always negated) + incrementState
        const float sd0 = fmaf(cpsi, r4, -(spsi * r5));
        const float sd1 = fmaf(spsi, r4, cpsi * r5);
        x = fmaf(sd0, a.dt, x);
        y = fmaf(sd1, a.dt, y);
        yaw = fmaf(-r6, a.dt, yaw);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t > 1) {  // finish step t-1: running mean over 1..T-1 (Q5); step 0 is never costed
        const int tp = t - 1;
        crash |= rc_p;
        const float c = cost_terms_b(a.cost, ct_p, tf_p, tb_p, crash);
        J = running_mean(J, c, tp, rt_p);
      }
      tf_p = tf; tb_p = tb; ct_p = ct; rc_p = rc; rt_p = rt;
    }
    // a hand-over that never arrived, in either wave: poison, do not hang (mppi_device.hpp)
    while (lds_peek(lds_addr(&fin[0])) == 0 && --budget > 0) __builtin_amdgcn_s_sleep(1);
    if (budget <= 0 || lds_peek(lds_addr(&fail[0])) != 0) J = __builtin_nanf("");
    a.costs[k] = J + 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// rollout_bf3_kernel: the two-wave form with the control work in a THIRD wavefront, like the control wave of
// the network kernels (rollout_mfma.hip): noise (the handle's MRG32k3a streams in the kernel, or the explicit
// eps buffer), the perturbed control, its write-back before the clamp (Q3), the clamp.  Nothing of
// mppi_controller.cu:136-153 depends on the state, so this wave runs up to kBfRing steps ahead, the dynamics
// wave touches no global memory in the T loop, and the stand-alone generator kernel in front of the rollout
// (7.9 us at K = 2560, T = 100) is gone.  Of the last step only the state record goes out (its update feeds
// nothing, mppi_controller.cu:160-177).  Same arithmetic, same order: bit-identical to the other two forms.
//   roles: 0 dynamics, 1 cost, 2 control (a.fault_wave == role + 1: mppi_debug_inject_handover_fault)
// ---------------------------------------------------------------------------------------------
// body of rollout_bf3_kernel for group `group` (64 rollouts) of instance `a`
__device__ __forceinline__ void bf3_group(const RolloutArgs &a, const int group)
{
  __shared__ __attribute__((aligned(16))) float W_s[4 * kNumBfs];  // transposed: [25][4]
  __shared__ float rec[kBfRing][8][kBfLanes];  // [slot][field][lane]: s3 s4 s5 s6 (dynamics) | u0 u1 du0 du1 (control)
  __shared__ int pub[kBfLanes], cpub[kBfLanes], done[kBfLanes], fail[4], fin[4];
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int i = threadIdx.x; i < 4 * kNumBfs; i += 3 * kBfLanes) W_s[(i % kNumBfs) * 4 + i / kNumBfs] = a.wpack[i];
  if (role == 0) { pub[lane] = 0; cpub[lane] = 0; done[lane] = 0; fail[lane & 3] = 0; fin[lane & 3] = 0; }
  __syncthreads();  // the only barrier
  const int k = group * kBfLanes + lane;
  const int K = a.K, T = a.T;
  const uint32_t a_pub = lds_addr(&pub[0]), a_cpub = lds_addr(&cpub[0]), a_done = lds_addr(&done[0]);
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == role + 1);  // mppi_device.hpp

  if (role == 0) {
    // ------------------------------ dynamics wave ------------------------------
    const uint32_t a_mypub = lds_addr(&pub[lane]);
    BfWeights Wr;
    Wr.load(W_s);
    float s[kStateDim];
#pragma unroll
    for (int i = 0; i < kStateDim; i++) s[i] = a.state[i];
    int seen = 0;  // steps the control wave has published
    for (int t = 0; t < T; t++) {
      // controls of step t: the control wave wrote them only after the cost wave had consumed step t - kBfRing,
      // so the slot's state fields are free as well
      while (seen < t + 1 && --budget > 0) seen = lds_peek(a_cpub);
      const int slot = t & (kBfRing - 1);
      const float u0 = rec[slot][4][lane], u1 = rec[slot][5][lane];
      rec[slot][0][lane] = s[3]; rec[slot][1][lane] = s[4]; rec[slot][2][lane] = s[5]; rec[slot][3][lane] = s[6];
      lds_publish(a_mypub, t + 1);
      if (t == T - 1) break;  // the last update feeds nothing
      float phi[kNumBfs], d[4];
      BasisShared c;
      basis_shared_fast(s, u0, c);
      basis_funcs_from(s, u1, c, phi);
      basis_dynamics_dev(Wr, phi, d);
#pragma unroll
      for (int i = 0; i < 4; i++) s[3 + i] = fmaf(d[i], a.dt, s[3 + i]);
    }
    spin_finish(budget, lds_addr(&fail[0]), lds_addr(&fin[0]));
  } else if (role == 2) {
    // ------------------------------ control wave ------------------------------
    const uint32_t a_mypub = lds_addr(&cpub[lane]);
    const bool inl = a.inline_noise != 0;
    Mrg gsta{0, 0, 0, 0, 0, 0};
    if (inl) {
      gsta.s10 = a.rng_in[k]; gsta.s11 = a.rng_in[K + k]; gsta.s12 = a.rng_in[2 * K + k];
      gsta.s20 = a.rng_in[3 * K + k]; gsta.s21 = a.rng_in[4 * K + k]; gsta.s22 = a.rng_in[5 * K + k];
    }
    float2 *const noise = reinterpret_cast<float2 *>(a.noise);
    const float2 *const Useq = reinterpret_cast<const float2 *>(a.U);
    const bool noise_free_k = (k == 0);      // mppi_controller.cu:136
    const bool pure_noise_k = (k >= a.k99);  // :141
    constexpr int kChunk = 4;  // steps of U / explicit eps requested at once
    int seen_d = 0, seen_c = 0;  // steps published by the dynamics wave / consumed by the cost wave
    for (int t0 = 0; t0 < T; t0 += kChunk) {
      float2 Uq[kChunk], eq[kChunk];
#pragma unroll
      for (int q = 0; q < kChunk; q++) {
        const int tq = min(t0 + q, T - 1);
        Uq[q] = Useq[tq];
        eq[q] = inl ? make_float2(0.0f, 0.0f) : noise[(size_t)tq * K + k];
      }
#pragma unroll
      for (int q = 0; q < kChunk; q++) {
        const int t = t0 + q;
        if (t < T) {
          const float2 e = inl ? noise_pair(gsta) : eq[q];
          float du0, du1, u0, u1;
          if (noise_free_k || t < a.opt_delay) {
            du0 = 0.0f; du1 = 0.0f; u0 = Uq[q].x; u1 = Uq[q].y;
          } else {
            du0 = e.x * a.nu[0];
            du1 = e.y * a.nu[1];
            u0 = pure_noise_k ? du0 : Uq[q].x + du0;
            u1 = pure_noise_k ? du1 : Uq[q].y + du1;
          }
          noise[(size_t)t * K + k] = make_float2(u0, u1);  // before the clamp (Q3)
          u0 = clampf(u0, a.u_lo[0], a.u_hi[0]);
          u1 = clampf(u1, a.u_lo[1], a.u_hi[1]);
          // slot t % kBfRing held step t - kBfRing: the dynamics wave read its controls before it published
          // that step, the cost wave is done with it once it has consumed it
          const int need = t - kBfRing + 1;
          while ((seen_d < need || seen_c < need) && --budget > 0) {
            seen_d = lds_peek(a_pub);
            seen_c = lds_peek(a_done);
            if (seen_d < need || seen_c < need) __builtin_amdgcn_s_sleep(2);
          }
          const int slot = t & (kBfRing - 1);
          rec[slot][4][lane] = u0; rec[slot][5][lane] = u1; rec[slot][6][lane] = du0; rec[slot][7][lane] = du1;
          lds_publish(a_mypub, t + 1);
        }
      }
    }
    if (inl) {
      a.rng_out[k] = gsta.s10; a.rng_out[K + k] = gsta.s11; a.rng_out[2 * K + k] = gsta.s12;
      a.rng_out[3 * K + k] = gsta.s20; a.rng_out[4 * K + k] = gsta.s21; a.rng_out[5 * K + k] = gsta.s22;
    }
    spin_finish(budget, lds_addr(&fail[0]), lds_addr(&fin[2]));
  } else {
    // -------------------------------- cost wave --------------------------------
    const uint32_t a_mydone = lds_addr(&done[lane]);
    float x = a.state[0], y = a.state[1], yaw = a.state[2];
    int crash = 0, seen = 0;
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
        while (seen < t + 1 && --budget > 0) {  // the dynamics wave publishes step t after it has read ctl(t)
          seen = lds_peek(a_pub);
          if (seen < t + 1) __builtin_amdgcn_s_sleep(1);
        }
        const int slot = t & (kBfRing - 1);
        const float r3 = rec[slot][0][lane], r4 = rec[slot][1][lane], r5 = rec[slot][2][lane], r6 = rec[slot][3][lane];
        const float u0 = rec[slot][4][lane], u1 = rec[slot][5][lane], du0 = rec[slot][6][lane], du1 = rec[slot][7][lane];
        lds_publish(a_mydone, t + 1);  // executes after the reads (the LDS runs a wave's instructions in order)
        rc = (int)((t > 0) & (fabsf(r3) >= kRollCrash));  // getCrash of update t-1
        float spsi, cpsi;
        sincos_fast(yaw, spsi, cpsi);
        const float st[3] = {x, y, yaw};
        if (t > 0) {
          if (a.cost.affine) track_fetch<true>(a.cost, st, cpsi, spsi, tf, tb);
          else track_fetch<false>(a.cost, st, cpsi, spsi, tf, tb);
        }
        if (a.cost.need_control_cost) cost_terms_a<true>(a.cost, a.nu, r4, r5, u0, u1, du0, du1, ct);
        else cost_terms_a<false>(a.cost, a.nu, r4, r5, u0, u1, du0, du1, ct);
        // computeKinematics (generalized_linear.cu:212-217, yaw rate always negated) + incrementState
        const float sd0 = fmaf(cpsi, r4, -(spsi * r5));
        const float sd1 = fmaf(spsi, r4, cpsi * r5);
        x = fmaf(sd0, a.dt, x);
        y = fmaf(sd1, a.dt, y);
        yaw = fmaf(-r6, a.dt, yaw);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t > 1) {  // finish step t-1: running mean over 1..T-1 (Q5); step 0 is never costed
        const int tp = t - 1;
        crash |= rc_p;
        const float c = cost_terms_b(a.cost, ct_p, tf_p, tb_p, crash);
        J = running_mean(J, c, tp, rt_p);
      }
      tf_p = tf; tb_p = tb; ct_p = ct; rc_p = rc; rt_p = rt;
    }
    // a hand-over that never arrived, in any wave: poison, do not hang (mppi_device.hpp)
    while ((lds_peek(lds_addr(&fin[0])) & lds_peek(lds_addr(&fin[2]))) == 0 && --budget > 0) __builtin_amdgcn_s_sleep(1);
    if (budget <= 0 || lds_peek(lds_addr(&fail[0])) != 0) J = __builtin_nanf("");
    a.costs[k] = J + 0.0f;
  }
}

__global__ __launch_bounds__(3 * kBfLanes) void rollout_bf3_kernel(const RolloutArgs a) { bf3_group(a, (int)blockIdx.x); }

// several instances in one launch (mppi_compute_control_batch: the two controllers of path_integral_bf's control
// loop): workgroup (x, y) runs group x of instance y
template <int NB>
__global__ __launch_bounds__(3 * kBfLanes) void rollout_bf3_batch_kernel(const QuadBatchArgsT<NB> b)
{
  // the instance from the workgroup's own index, its argument block at a compile-time position (MPPI_BATCH_DISPATCH)
#define MPPI_BF_BODY(A)                                  \
  do {                                                   \
    if ((int)blockIdx.x >= (A).K / kBfLanes) return;     \
    bf3_group((A), (int)blockIdx.x);                     \
  } while (0)
  MPPI_BATCH_DISPATCH(NB, b, MPPI_BF_BODY);
#undef MPPI_BF_BODY
}

// test entry (mppi_debug_dynamics): state derivative of n independent (state, control) pairs
__global__ __launch_bounds__(kBfLanes) void dynamics_bf_kernel(const float *W, const float *states,
                                                               const float *controls, float *ders, int n)
{
  __shared__ __attribute__((aligned(16))) float W_s[4 * kNumBfs];  // transposed: [25][4]
  const int lane = threadIdx.x;
  for (int i = lane; i < 4 * kNumBfs; i += kBfLanes) W_s[(i % kNumBfs) * 4 + i / kNumBfs] = W[i];
  __syncthreads();
  const int idx = blockIdx.x * kBfLanes + lane;
  if (idx >= n) return;
  float s[kStateDim];
#pragma unroll
  for (int i = 0; i < kStateDim; i++) s[i] = states[idx * kStateDim + i];
  float spsi, cpsi;
  sincos_fast(s[2], spsi, cpsi);
  float sd[kStateDim];
  BfWeights Wr;
  Wr.load(W_s);
  bf_state_deriv(Wr, s, controls[idx * 2], controls[idx * 2 + 1], cpsi, spsi, sd);
#pragma unroll
  for (int i = 0; i < kStateDim; i++) ders[idx * kStateDim + i] = sd[i];
}

hipError_t launch_rollout_bf(const RolloutArgs &a, int waves, hipStream_t stream)
{
  if (waves == 3) MPPI_LAUNCH_ROLLOUT(rollout_bf3_kernel, dim3(a.K / kBfLanes), dim3(3 * kBfLanes), 0, stream, a);
  else if (waves == 2) MPPI_LAUNCH_ROLLOUT(rollout_bf2_kernel, dim3(a.K / kBfLanes), dim3(2 * kBfLanes), 0, stream, a);
  else MPPI_LAUNCH_ROLLOUT(rollout_bf_kernel, dim3(a.K / kBfLanes), dim3(kBfLanes), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_rollout_bf_batch(const QuadBatchArgs &b, hipStream_t stream)
{
  if (b.n < 1 || b.n > kMaxBatch) return hipErrorInvalidValue;
  int gmax = 0;
  for (int i = 0; i < b.n; i++) gmax = b.inst[i].K / kBfLanes > gmax ? b.inst[i].K / kBfLanes : gmax;
  if (b.n <= 2) hipLaunchKernelGGL(rollout_bf3_batch_kernel<2>, dim3(gmax, b.n), dim3(3 * kBfLanes), 0, stream, batch_args_prefix<2>(b));
  else hipLaunchKernelGGL(rollout_bf3_batch_kernel<4>, dim3(gmax, b.n), dim3(3 * kBfLanes), 0, stream, b);
  return hipGetLastError();
}

hipError_t launch_dynamics_bf(const float *W, const float *states, const float *controls, float *ders, int n,
                              hipStream_t stream)
{
  hipLaunchKernelGGL(dynamics_bf_kernel, dim3((n + kBfLanes - 1) / kBfLanes), dim3(kBfLanes), 0, stream, W, states,
                     controls, ders, n);
  return hipGetLastError();
}

}  // namespace mppi
