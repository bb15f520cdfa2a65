// rollout_row.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) for gfx950, the LATENCY form of 32-wide nets:
// the network on the VECTOR ALU, four rollouts per dynamics wavefront, four dynamics wavefronts + the four riders of
// group_roles.hpp (pose -> cost, noise -> control) per 16 rollouts.  Used while every group has a CU of its own
// (K <= 16 x #CUs: BASELINE configs 1-3, the reference's K = 1920).
//
// Why not the matrix instruction here: at one group per CU the T-step recurrence is a latency chain, and a dependent
// k-step of v_mfma_f32_16x16x4_f32 costs ~8 cycles (4 k per 32-cycle instruction), so a 32-input layer
// is 8 x 33 = 264 cycles on top of a hand-over between the two waves that share the layer (the quad form: ~1 530
// cycles per step).  A dependent v_pk_fma_f32 also issues every 8 cycles but carries TWO neurons and needs no partner:
//   * one dynamics wave = 4 rollouts x 16 lanes; lane (r, p) owns neurons 2p, 2p+1 of every hidden layer of rollout r
//     (outputs 2(p&1), 2(p&1)+1 of the last layer), their weights in registers as pairs;
//   * a layer = per lane the k-ascending fmaf chain of mppi_controller.cu's dot product (neural_net_model.cu:379-394;
//     bias afterwards) -- bit-identical to every other form -- with the activation a_k broadcast to both halves of the
//     packed multiply-add (op_sel);
//   * the activations of a layer go through LDS inside the wave: one 8-B write per lane, eight 16-B reads per lane that
//     are broadcasts within the 16 lanes of a rollout.  No other wave is involved: no sequence word, no poll, no
//     barrier on the recurrence (tools/ub/row_lds_ub.hip: 1 154 cycles per step alone on a SIMD);
//   * every lane pair (p, p ^ 1) computes the same two outputs of the last layer, so the new state reaches layer 0 of
//     the next step through one DPP move instead of a third LDS round trip;
//   * state records, controls, texels, noise: the rings and riders of group_roles.hpp, one rider per SIMD beside one
//     dynamics wave.
#include "group_roles.hpp"
#include "mppi_kernels.hpp"

namespace mppi {

// Diagnostic build only (-DMPPI_ROW_STAMPS, tools/row_stamps.py): s_memtime stamps of workgroup 0 -- where the time of a
// launch goes outside the T loop.  The product build has no stamp instruction.
#ifdef MPPI_ROW_STAMPS
__device__ unsigned long long g_row_stamps[16];
#define RSTAMP(i)                                                                                         \
  do {                                                                                                    \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                                     \
      unsigned long long t__;                                                                             \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                           \
      g_row_stamps[i] = t__;                                                                              \
    }                                                                                                     \
  } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif

template <int H>
struct RowShared {
  static constexpr int NW = 4;            // dynamics waves per group, four rollouts each
  static constexpr int NSW = 1;           // xseq[w] = steps published by dynamics wave w
  static constexpr bool kRecByAll = true; // every dynamics wave writes the state records of its own rollouts
  int xseq[NW][64];
  float rec[kGRing][kRolloutsPerWave][4];   // s3..s6 before the update of step t; also the layer-0 input of that step
  int cost_done[64];
  float ctl_b1[kGRing][64];
  float ctl_rec[kGRing][kRolloutsPerWave][4];
  int ctl_pub[64];
  float tex[kGRing][kRolloutsPerWave][2];
  int pose_pub[64];
  float eps[kGRing][kRolloutsPerWave][2];
  int rng_pub[64];
  int fail[4];
  int fin[8];
  float act[NW][2][4][H];                   // per dynamics wave: activations of layer 0 / layer 1 of its four rollouts
};

template <int H>
struct RowWeights {
  f32x2 w1[kNetIn], w2[H], w3[H];
  f32x2 b1s, b2s, b3;  // hidden biases pre-scaled for tanh_bias2 (theta_s of the register VALU kernel)
};

// rowpack: the weights in REGISTER order, written by the host (pack_row_weights, mppi_abi.hip): 16-B entry i of lane p at
// float4 index i * 16 + p -- a load instruction of a wave reads 256 contiguous bytes (the four rollouts of a wave share
// them).  Entries: 0..2 = w1[0..5], 3..18 = w2[0..31], 19..34 = w3[0..31] (pairs, two per entry), 35 = (b1s, b2s), 36 = b3.
// (Loading the rows straight from the packed theta -- 44 scattered 16-B loads per lane -- cost 2.1 us per launch.)
constexpr int kRowPackEntries = 37;
template <int H>
__device__ __forceinline__ void row_load(const float *rowpack, int p, RowWeights<H> &W)
{
  static_assert(H == 32, "entry layout of pack_row_weights");
  const float4 *pk = reinterpret_cast<const float4 *>(rowpack) + p;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float4 v = pk[i * 16];
    W.w1[2 * i] = f32x2{v.x, v.y};
    W.w1[2 * i + 1] = f32x2{v.z, v.w};
  }
#pragma unroll
  for (int i = 0; i < H / 2; i++) {
    const float4 v = pk[(3 + i) * 16], u = pk[(3 + H / 2 + i) * 16];
    W.w2[2 * i] = f32x2{v.x, v.y};
    W.w2[2 * i + 1] = f32x2{v.z, v.w};
    W.w3[2 * i] = f32x2{u.x, u.y};
    W.w3[2 * i + 1] = f32x2{u.z, u.w};
  }
  const float4 b = pk[35 * 16], c = pk[36 * 16];
  W.b1s = f32x2{b.x, b.y};
  W.b2s = f32x2{b.z, b.w};
  W.b3 = f32x2{c.x, c.y};
}

// z = sum_k w[k] * a[k], k ascending, one fmaf per k and neuron (the pair shares a[k])
template <int N>
__device__ __forceinline__ f32x2 row_dot(const f32x2 (&w)[N], const float4 (&v)[N / 4])
{
  f32x2 z = {0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < N / 4; q++) {
    z = __builtin_elementwise_fma(w[4 * q + 0], f32x2{v[q].x, v[q].x}, z);
    z = __builtin_elementwise_fma(w[4 * q + 1], f32x2{v[q].y, v[q].y}, z);
    z = __builtin_elementwise_fma(w[4 * q + 2], f32x2{v[q].z, v[q].z}, z);
    z = __builtin_elementwise_fma(w[4 * q + 3], f32x2{v[q].w, v[q].w}, z);
  }
  return z;
}

// the partner lane's pair (lane p ^ 1 of the same rollout): one DPP move per register, no LDS
__device__ __forceinline__ f32x2 row_partner(f32x2 v)
{
  f32x2 o;
  o.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.x), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
  o.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.y), 0xB1, 0xF, 0xF, false));
  return o;
}

template <int H>
__device__ __forceinline__ void row_dynamics(const RolloutArgs &a, RowShared<H> &sh, const int w)
{
  const int lane = threadIdx.x & 63;
  const int r = lane >> 4, p = lane & 15;
  const int jr = 4 * w + r;  // rollout of the group
  const bool odd = (p & 1) != 0;
  const int T = a.T;
  RowWeights<H> W;
#ifdef MPPI_DIAG_NOWLOAD  // diagnostic build: what do the weight loads cost at the start of a launch?
  for (int k = 0; k < kNetIn; k++) W.w1[k] = f32x2{0.01f * k, 0.02f};
  for (int k = 0; k < H; k++) { W.w2[k] = f32x2{0.001f * k, 0.002f * p}; W.w3[k] = f32x2{0.003f, 0.001f * k}; }
  W.b1s = W.b2s = W.b3 = f32x2{0.01f, 0.02f};
#else
  row_load<H>(a.wpack, p, W);
#endif
  // pinned: the waits for the weight loads sit here, not at their first use inside the T loop
#pragma unroll
  for (int k = 0; k < H; k++) { asm volatile("" : "+v"(W.w2[k])); asm volatile("" : "+v"(W.w3[k])); }

  const uint32_t a_myseq = lds_addr(&sh.xseq[w][lane]);
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile float __attribute__((address_space(3))) *lds_float_p;
  const lds_int_p p_pub = (lds_int_p)&sh.ctl_pub[0];
  const lds_int_p p_cd = (lds_int_p)&sh.cost_done[0];
  const lds_float_p p_u = (lds_float_p)&sh.ctl_rec[0][jr][0];  // clamped u0, u1 of this lane's rollout (control wave)
  constexpr int kCtlSlot = kRolloutsPerWave * 4;                // floats per ring slot of ctl_rec
  float(*act0)[H] = sh.act[w][0];
  float(*act1)[H] = sh.act[w][1];

  // this lane's pair of the state: (s3, s4) for even p, (s5, s6) for odd p -- every lane pair (p, p ^ 1) of a rollout
  // computes the same output pair, so the whole state is one DPP move away
  f32x2 sp = odd ? f32x2{a.state[5], a.state[6]} : f32x2{a.state[3], a.state[4]};
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == w + 1);
  if (w == 0) RSTAMP(2);  // weights in registers
  while (__builtin_amdgcn_readfirstlane(*p_pub) < 1 && --budget > 0) __builtin_amdgcn_s_sleep(1);
  if (w == 0) RSTAMP(3);  // first controls published: the T loop starts
  float u0n = p_u[0], u1n = p_u[1];
  asm volatile("" : "+v"(u0n), "+v"(u1n));  // pinned: the wait for these reads sits here, not inside the loop

  // Steps 0 .. T-2 in full; of step T-1 only the state record goes out (its update feeds nothing: the cost is the
  // running mean over the states BEFORE the updates of steps 1..T-1, mppi_controller.cu:160-177)
  for (int t = 0; t < T - 1; t++) {
    const int slot = t & (kGRing - 1);
    const float u0 = u0n, u1 = u1n;
    const f32x2 so = row_partner(sp);
    const f32x2 slo = odd ? so : sp, shi = odd ? sp : so;  // (s3, s4), (s5, s6)
    // record for the pose / cost waves: the state BEFORE the update (the ring slot was checked at the end of the
    // previous step); then the publication -- which also says: this wave is done with the control record of step t
    if (p < 2) *reinterpret_cast<f32x2 *>(&sh.rec[slot][jr][2 * p]) = sp;
    lds_publish(a_myseq, t + 1);
    // layer 0: [s3, s4, s5, s6, u0, u1]
    f32x2 z = {0.0f, 0.0f};
    z = __builtin_elementwise_fma(W.w1[0], f32x2{slo.x, slo.x}, z);
    z = __builtin_elementwise_fma(W.w1[1], f32x2{slo.y, slo.y}, z);
    z = __builtin_elementwise_fma(W.w1[2], f32x2{shi.x, shi.x}, z);
    z = __builtin_elementwise_fma(W.w1[3], f32x2{shi.y, shi.y}, z);
    z = __builtin_elementwise_fma(W.w1[4], f32x2{u0, u0}, z);
    z = __builtin_elementwise_fma(W.w1[5], f32x2{u1, u1}, z);
    *reinterpret_cast<f32x2 *>(&act0[r][2 * p]) = tanh_bias2(z, W.b1s);
    __builtin_amdgcn_wave_barrier();
    {
      float4 v[H / 4];
#pragma unroll
      for (int q = 0; q < H / 4; q++) v[q] = *reinterpret_cast<const float4 *>(&act0[r][4 * q]);
      *reinterpret_cast<f32x2 *>(&act1[r][2 * p]) = tanh_bias2(row_dot<H>(W.w2, v), W.b2s);
    }
    __builtin_amdgcn_wave_barrier();
    // requested now, used at the end of the step (behind the output layer): the control wave's count, then this
    // rollout's controls of step t+1 (valid if the count read before them is >= t+2), and the cost wave's progress
    const int sn = ((t + 1) & (kGRing - 1)) * kCtlSlot;
    const int cp_v = *p_pub;
    float un0_v = p_u[sn], un1_v = p_u[sn + 1];
    const int cd_v = *p_cd;
    {
      float4 v[H / 4];
#pragma unroll
      for (int q = 0; q < H / 4; q++) v[q] = *reinterpret_cast<const float4 *>(&act1[r][4 * q]);
      const f32x2 d = row_dot<H>(W.w3, v) + W.b3;
      sp = __builtin_elementwise_fma(d, f32x2{a.dt, a.dt}, sp);  // incrementState, neural_net_model.cu:334-344
    }
    // step t+1 may start when the control wave has published it (it runs ahead) and the ring slot of its state record
    // is free: that slot held step t+1 - kGRing, consumed once cost_done >= t+2 - kGRing
    // (a shorter leash for the riders -- waiting when the cost wave is more than 2 / 3 / 5 steps behind instead of a full
    // ring -- was measured: 124 / 72.6 / 58.0 us against 55.2 us; the riders need the slack)
    const int want = t + 2, want_cd = t + 2 - kGRing;
    int cp = __builtin_amdgcn_readfirstlane(cp_v), cd = __builtin_amdgcn_readfirstlane(cd_v);
    while (((cp < want) | (cd < want_cd)) && --budget > 0) {
      cp = __builtin_amdgcn_readfirstlane(*p_pub);
      un0_v = p_u[sn];
      un1_v = p_u[sn + 1];
      cd = __builtin_amdgcn_readfirstlane(*p_cd);
    }
    u0n = un0_v;
    u1n = un1_v;
  }
  if (w == 0) RSTAMP(4);  // T loop done
  {  // the record of step T-1
    const int t = T - 1;
    if (p < 2) *reinterpret_cast<f32x2 *>(&sh.rec[t & (kGRing - 1)][jr][2 * p]) = sp;
    __builtin_amdgcn_wave_barrier();
    lds_publish(a_myseq, t + 1);
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[w]));
}

template <int H, bool AFFINE, bool CTRL>
__global__ __launch_bounds__(512) void rollout_row_kernel(const RolloutArgs a)
{
  using SH = RowShared<H>;
  using R = GroupRoles<SH>;
  __shared__ __attribute__((aligned(16))) SH sh;
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (role == 0) RSTAMP(0);  // first instruction
  if (role == 0) {  // sequence words start at 0; the only barrier
#pragma unroll
    for (int w = 0; w < 4; w++) sh.xseq[w][lane] = 0;
    sh.cost_done[lane] = 0;
    sh.ctl_pub[lane] = 0;
    sh.pose_pub[lane] = 0;
    sh.rng_pub[lane] = 0;
    sh.fail[lane & 3] = 0;
    sh.fin[lane & 7] = 0;
  }
  __syncthreads();
  if (role == 0) RSTAMP(1);  // behind the barrier
#ifdef MPPI_ROW_RIDER_PRIO
  if (role >= 4) __builtin_amdgcn_s_setprio(MPPI_ROW_RIDER_PRIO);
#endif
  if (role < 4) row_dynamics<H>(a, sh, role);
  else if (role == R::kCost) { group_cost_wave<SH, CTRL>(a, sh); RSTAMP(5); }  // costs stored
  else if (role == R::kCtl) { group_control_wave(a, sh); RSTAMP(6); }
  else if (role == R::kPose) { group_pose_wave<SH, AFFINE>(a, sh); RSTAMP(7); }
  else { group_rng_wave(a, sh); RSTAMP(8); }
}

// several instances in one launch (mppi_compute_control_batch): workgroups [first[i], first[i+1]) run instance i, whose
// argument block carries group0 = first[i]
template <int H, bool AFFINE, bool CTRL>
__global__ __launch_bounds__(512) void rollout_row_batch_kernel(const QuadBatchArgs b)
{
  using SH = RowShared<H>;
  using R = GroupRoles<SH>;
  __shared__ __attribute__((aligned(16))) SH sh;
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int i = 0;  // workgroup-uniform
#pragma unroll
  for (int q = 1; q < kMaxBatch; q++)
    if (q < b.n && (int)blockIdx.x >= b.first[q]) i = q;
  const RolloutArgs &a = b.inst[i];
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
  __syncthreads();
  if (role < 4) row_dynamics<H>(a, sh, role);
  else if (role == R::kCost) group_cost_wave<SH, CTRL>(a, sh);
  else if (role == R::kCtl) group_control_wave(a, sh);
  else if (role == R::kPose) group_pose_wave<SH, AFFINE>(a, sh);
  else group_rng_wave(a, sh);
}

bool row_variant_supported(int hidden, int n_hidden) { return hidden == 32 && n_hidden == 2; }
int row_pack_floats() { return kRowPackEntries * 16 * 4; }

hipError_t launch_rollout_row_batch(const QuadBatchArgs &b, hipStream_t stream)
{
  if (b.n < 1 || b.n > kMaxBatch) return hipErrorInvalidValue;
  bool affine = true, ctrl = false;  // the general forms are exact supersets (rollout_mfma.hip)
  for (int i = 0; i < b.n; i++) {
    affine = affine && b.inst[i].cost.affine != 0;
    ctrl = ctrl || b.inst[i].cost.need_control_cost != 0;
  }
  const dim3 grid(b.first[b.n]), block(512);
  if (affine && !ctrl) hipLaunchKernelGGL((rollout_row_batch_kernel<32, true, false>), grid, block, 0, stream, b);
  else if (affine && ctrl) hipLaunchKernelGGL((rollout_row_batch_kernel<32, true, true>), grid, block, 0, stream, b);
  else if (!affine && !ctrl) hipLaunchKernelGGL((rollout_row_batch_kernel<32, false, false>), grid, block, 0, stream, b);
  else hipLaunchKernelGGL((rollout_row_batch_kernel<32, false, true>), grid, block, 0, stream, b);
  return hipGetLastError();
}

hipError_t launch_rollout_row(int hidden, int n_hidden, const RolloutArgs &a, hipStream_t stream)
{
  if (!row_variant_supported(hidden, n_hidden) || a.K % kRolloutsPerWave != 0) return hipErrorInvalidValue;
  const bool affine = a.cost.affine != 0, ctrl = a.cost.need_control_cost != 0;
  const dim3 grid(a.K / kRolloutsPerWave), block(512);
  if (affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_row_kernel<32, true, false>), grid, block, 0, stream, a);
  else if (affine && ctrl) MPPI_LAUNCH_ROLLOUT((rollout_row_kernel<32, true, true>), grid, block, 0, stream, a);
  else if (!affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_row_kernel<32, false, false>), grid, block, 0, stream, a);
  else MPPI_LAUNCH_ROLLOUT((rollout_row_kernel<32, false, true>), grid, block, 0, stream, a);
  return hipGetLastError();
}

}  // namespace mppi

#ifdef MPPI_ROW_STAMPS
extern "C" int mppi_debug_read_row_stamps(unsigned long long *out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mppi::g_row_stamps), sizeof(unsigned long long) * 16);
}
#endif
