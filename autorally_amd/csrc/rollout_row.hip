// rollout_row.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) for gfx950, the LATENCY form of 32-wide nets:
// the network on the VECTOR ALU, four rollouts per dynamics wavefront, four dynamics wavefronts + the four riders of
// group_roles.hpp (pose -> cost, noise -> control) per 16 rollouts.  Used while every group has a CU of its own
// (K <= 16 x #CUs: BASELINE configs 1-3, the reference's K = 1920).
//
// Why not the matrix instruction here: at one group per CU the T-step recurrence is a latency chain, and a dependent
// k-step of v_mfma_f32_16x16x4_f32 costs ~8 cycles (4 k per 32-cycle instruction), so a 32-input layer
// is 8 x 33 = 264 cycles on top of a hand-over between the two waves that share the layer (the quad form: ~1 530
// cycles per step).  A dependent v_pk_fma_f32 issues every ~9 cycles too but carries TWO neurons and needs no partner:
//   * one dynamics wave = 4 rollouts x 16 lanes = 4 DPP ROWS; lane (r, p) owns neurons 2p, 2p+1 of every hidden layer of
//     rollout r (outputs 2(p&1), 2(p&1)+1 of the last layer), their weights in registers as pairs;
//   * a layer = per lane the k-ascending fmaf chain of mppi_controller.cu's dot product (neural_net_model.cu:379-394;
//     bias afterwards) -- bit-identical to every other form -- with the activation a_k broadcast to both halves of the
//     packed multiply-add (op_sel);
//   * the activations of a layer never leave the registers: `row_newbcast:q` of the DPP hands every lane of a row the
//     value of lane q, one v_mov_b64_dpp per PAIR of k (round 4; one v_mov_b32_dpp per k before), issued in the shadow of
//     the multiply-adds (round 3, second half;
//     before that a layer's activations went through LDS -- one 8-B write and eight 16-B broadcast reads per lane and layer:
//     tools/ub/row_bcast_ub.hip measures the recurrence alone on a SIMD at 865 cycles per step against 1 120).  No other
//     wave is involved: no sequence word, no poll, no barrier on the recurrence;
//   * lanes 0 and 1 of a row hold the state pair (s3, s4) / (s5, s6) (every even / odd lane computes the same two
//     outputs), so layer 0 of the next step reads the new state through four of the same moves;
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
  static constexpr int kR = 16;           // rollouts per group
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
  float gstate[8];  // gated launch: the vehicle state the pose wave took from the gate block, then 1 in gate_open[]
  int gate_open[8];
  float dump[NW][64 * 2 + (kGRing - 1) * kRolloutsPerWave * 4];  // where lanes p >= 2 of a dynamics wave put their copy of the state
                                                                 // pair (never read): 8 B per lane, moved along with the ring slot
};

// TREE (the "row_tree" form): the output layer as own-activation partials + a butterfly over the row instead of the
// k-ascending chain -- see row_out_tree below; w3 then holds this lane's 2 x 4 output weights instead of all 32 x 2.
template <int H, bool TREE>
struct RowWeights {
  f32x2 w1[kNetIn], w2[H], w3[TREE ? 4 : H];
  f32x2 b1s, b2s, b3;  // hidden biases pre-scaled for tanh_bias2 (theta_s of the register VALU kernel)
};

// rowpack: the weights in REGISTER order, written by the host (pack_row_weights, mppi_abi.hip): 16-B entry i of lane p at
// float4 index i * 16 + p -- a load instruction of a wave reads 256 contiguous bytes (the four rollouts of a wave share
// them).  Entries: 0..2 = w1[0..5], 3..18 = w2[0..31], 19..34 = w3[0..31] (pairs, two per entry), 35 = (b1s, b2s), 36 = b3.
// Tree form: 37 = (W3[o][2p], W3[o^1][2p], W3[o][2p+1], W3[o^1][2p+1]), 38 = the same of outputs o^2, o^3, 39 = (b3[o], -, -, -),
// o = p >> 2 the output this lane's quad ends up with.
// (Loading the rows straight from the packed theta -- 44 scattered 16-B loads per lane -- cost 2.1 us per launch.)
constexpr int kRowPackEntries = 40;
template <int H, bool TREE>
__device__ __forceinline__ void row_load(const float *rowpack, int p, RowWeights<H, TREE> &W)
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
    const float4 v = pk[(3 + i) * 16];
    W.w2[2 * i] = f32x2{v.x, v.y};
    W.w2[2 * i + 1] = f32x2{v.z, v.w};
    if constexpr (!TREE) {
      const float4 u = pk[(3 + H / 2 + i) * 16];
      W.w3[2 * i] = f32x2{u.x, u.y};
      W.w3[2 * i + 1] = f32x2{u.z, u.w};
    }
  }
  const float4 b = pk[35 * 16];
  W.b1s = f32x2{b.x, b.y};
  W.b2s = f32x2{b.z, b.w};
  if constexpr (TREE) {
    const float4 u = pk[37 * 16], v = pk[38 * 16], c = pk[39 * 16];
    W.w3[0] = f32x2{u.x, u.y};
    W.w3[1] = f32x2{u.z, u.w};
    W.w3[2] = f32x2{v.x, v.y};
    W.w3[3] = f32x2{v.z, v.w};
    W.b3 = f32x2{c.x, c.y};
  } else {
    const float4 c = pk[36 * 16];
    W.b3 = f32x2{c.x, c.y};
  }
}

// A value of this lane's rollout from the registers of the lane that holds it: a rollout is one 16-lane DPP row,
// `row_newbcast:q` hands every lane of a row the value of lane q (off the dependent chain).  gfx90a+ DPP control 0x150 + q;
// works on 32-bit operands on gfx950 (tools/ub/row_bcast_ub.hip checks the bits against the LDS form) and on 64-bit ones
// (row_bc2 below).  This 32-bit form brings the state components to layer 0.
template <int Q>
__device__ __forceinline__ float row_bc(float a)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x150 + Q, 0xF, 0xF, false));
}
// z = sum_k w[k] * a[k] over the 32 activations of the rollout (lane q holds a[2q], a[2q+1]), k ascending: ONE chain of
// dependent v_pk_fma_f32 (8.5 cycles apart, tools/ub/valu_issue_ub.hip).  The PAIR of lane q comes by one v_mov_b64_dpp
// `row_newbcast:q` (the 64-bit move takes the same DPP control) and serves the multiply-adds of k = 2q (both halves read
// the pair's low word: op_sel) and k = 2q+1 (the high word): 16 moves + 32 multiply-adds per layer instead of 32 + 32 with
// 32-bit moves -- the same arithmetic bit for bit, rollout 36.9 -> 34.3 us, step 0.0486 -> 0.0461 ms
// (profiles/r04_r_mov64_ab.txt).  Written as the schedule it has to be -- multiply-add of 2q, the move for q+1 in its
// shadow, multiply-add of 2q+1 -- and held there by scheduling barriers (left alone inside the kernel, the scheduler hoists
// all moves in front of the chain: ~130 cycles more per layer).
template <int Q>
__device__ __forceinline__ f32x2 row_bc2(f32x2 a)
{
  return __builtin_bit_cast(f32x2, (long long)__builtin_amdgcn_mov_dpp(__builtin_bit_cast(long long, a), 0x150 + Q, 0xF, 0xF, false));
}
template <int Q>
__device__ __forceinline__ void row_dot_step(f32x2 &z, f32x2 &b, const f32x2 *w, f32x2 a)
{
  z = __builtin_elementwise_fma(w[2 * Q], f32x2{b.x, b.x}, z);
  __builtin_amdgcn_sched_barrier(0);
  const f32x2 bn = row_bc2<(Q + 1 < 16 ? Q + 1 : 15)>(a);
  __builtin_amdgcn_sched_barrier(0);
  z = __builtin_elementwise_fma(w[2 * Q + 1], f32x2{b.y, b.y}, z);
  __builtin_amdgcn_sched_barrier(0);
  b = bn;
}
__device__ __forceinline__ f32x2 row_dot_bc(const f32x2 *w, f32x2 a)
{
  f32x2 z = {0.0f, 0.0f};
  f32x2 b = row_bc2<0>(a);
  __builtin_amdgcn_sched_barrier(0);
#define RD4(Q) row_dot_step<Q>(z, b, w, a); row_dot_step<Q + 1>(z, b, w, a); row_dot_step<Q + 2>(z, b, w, a); row_dot_step<Q + 3>(z, b, w, a);
  RD4(0) RD4(4) RD4(8) RD4(12)
#undef RD4
  return z;
}

// The output layer of the TREE form ("row_tree").  In the exact form every lane of a rollout runs the whole 32-long chain
// for two of the four outputs -- 32 dependent packed multiply-adds and 32 moves per lane and step, ~290 of the step's ~930
// cycles and 30 % of the kernel's vector instructions, for 4 useful numbers.  Here lane p multiplies only ITS OWN two
// activations (a[2p], a[2p+1]) into the four outputs (a product and a fused multiply-add each, packed: 4 instructions) and
// the 16 partials of an output are summed by a butterfly of DPP adds that halves the number of live values per level:
//   level 1 (row_ror:8):        v0 += v2(p^8), v1 += v3(p^8)    lanes p < 8 keep outputs {0,1}, lanes p >= 8 outputs {2,3}
//   level 2 (row_half_mirror):  v0 += v1(p^7)                   quad q = p >> 2 keeps output q
//   level 3, 4 (quad_perm):     v0 += v0(p^1); v0 += v0(p^2)
// -- 5 adds on a 4-deep chain; which output a lane keeps is wired into the ORDER of its weights (v_i = partial of output
// (p >> 2) ^ i, pack_row_weights), so no select is needed.  Every lane of quad q ends with output q = state component
// s[3 + q]: layer 0 of the next step takes the state from lanes 0, 4, 8, 12 of the row.  This is NOT the reference's
// summation order (neural_net_model.cu:379-394 sums k ascending; the hidden layers keep that order): the form is opt-in by
// tolerance -- checked bit for bit against the test oracle's mode 2 (its out_tree_dot), and against the
// nominal oracle at the north-star criteria (controls 1e-4).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float acc, float src)
{
  return acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_out_tree(const f32x2 *w3, f32x2 a)
{
  const f32x2 ax = {a.x, a.x}, ay = {a.y, a.y};
  const f32x2 v01 = __builtin_elementwise_fma(w3[1], ay, w3[0] * ax);
  const f32x2 v23 = __builtin_elementwise_fma(w3[3], ay, w3[2] * ax);
  float v0 = dpp_add<0x128>(v01.x, v23.x);  // row_ror:8
  const float v1 = dpp_add<0x128>(v01.y, v23.y);
  v0 = dpp_add<0x141>(v0, v1);              // row_half_mirror: lane p <- lane p ^ 7
  v0 = dpp_add<0xB1>(v0, v0);               // quad_perm [1,0,3,2]
  v0 = dpp_add<0x4E>(v0, v0);               // quad_perm [2,3,0,1]
  return v0;
}

template <int H, bool TREE, bool GATED = false>
__device__ __forceinline__ void row_dynamics(const RolloutArgs &a, RowShared<H> &sh, const int w)
{
  static_assert(H == 32, "row_dot_bc: 32 activations, two per lane of a 16-lane row");
  const int lane = threadIdx.x & 63;
  const int r = lane >> 4, p = lane & 15;
  const int jr = 4 * w + r;  // rollout of the group
  const bool odd = (p & 1) != 0;
  const int T = a.T;
  RowWeights<H, TREE> W;
  row_load<H, TREE>(a.wpack, p, W);
  // pinned: the waits for the weight loads sit here, not at their first use inside the T loop
#pragma unroll
  for (int k = 0; k < H; k++) asm volatile("" : "+v"(W.w2[k]));
#pragma unroll
  for (int k = 0; k < (TREE ? 4 : H); k++) asm volatile("" : "+v"(W.w3[k]));

  const uint32_t a_myseq = lds_addr(&sh.xseq[w][lane]);
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile f32x2 __attribute__((address_space(3))) *lds_f2_p;
  const lds_int_p p_pub = (lds_int_p)&sh.ctl_pub[0];
  const lds_f2_p p_u = (lds_f2_p)&sh.ctl_rec[0][jr][0];  // clamped (u0, u1) of this lane's rollout (control wave), ring slot 0
  constexpr int kSlotF2 = kRolloutsPerWave * 2;          // f32x2 per ring slot of ctl_rec (and of rec)
  // The state record of a step is stored by EVERY lane: lanes 0, 1 of a row into the record, the others into a dump row
  // nobody reads -- no exec masking on the recurrence; both move along with the ring slot (one address add for all lanes).
  // (tree form: one component per lane -- quad q of the row holds s[3 + q], lanes 0, 4, 8, 12 write the record)
  const uint32_t a_rec0 = TREE ? (((p & 3) == 0) ? lds_addr(&sh.rec[0][jr][p >> 2]) : lds_addr(&sh.dump[w][2 * lane]))
                               : ((p < 2) ? lds_addr(&sh.rec[0][jr][2 * p]) : lds_addr(&sh.dump[w][2 * lane]));
  constexpr uint32_t kRecStride = sizeof(float) * kRolloutsPerWave * 4;

  // this lane's pair of the state: (s3, s4) for even p, (s5, s6) for odd p -- every even / odd lane of a rollout computes
  // the same output pair; layer 0 takes the pairs of lanes 0 and 1 of the row
  // (tree form: sp.x = s[3 + (p >> 2)], sp.y unused)
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == w + 1);
  f32x2 sp;
  if constexpr (GATED) {
    // the state arrives through the gate block: the pose wave has put it into LDS (the weights above were loaded meanwhile)
    const uint32_t a_go = lds_addr(&sh.gate_open[0]);
    while (lds_peek(a_go) == 0 && --budget > 0) __builtin_amdgcn_s_sleep(1);
    const volatile float *gs = sh.gstate;
    sp = TREE ? f32x2{gs[3 + (p >> 2)], 0.0f} : odd ? f32x2{gs[5], gs[6]} : f32x2{gs[3], gs[4]};
  } else {
    sp = TREE ? f32x2{a.state[3 + (p >> 2)], 0.0f} : odd ? f32x2{a.state[5], a.state[6]} : f32x2{a.state[3], a.state[4]};
  }
  if (w == 0) RSTAMP(2);  // weights in registers
  while (__builtin_amdgcn_readfirstlane(*p_pub) < 1 && --budget > 0) __builtin_amdgcn_s_sleep(1);
  if (w == 0) RSTAMP(3);  // first controls published: the T loop starts
  f32x2 un = p_u[0];
  asm volatile("" : "+v"(un));  // pinned: the wait for this read sits here, not inside the loop

  // Steps 0 .. T-2 in full; of step T-1 only the state record goes out (its update feeds nothing: the cost is the
  // running mean over the states BEFORE the updates of steps 1..T-1, mppi_controller.cu:160-177)
  for (int t = 0; t < T - 1; t++) {
    const int slot = t & (kGRing - 1);
    const f32x2 u = un;
    const f32x2 slo = TREE ? f32x2{row_bc<0>(sp.x), row_bc<4>(sp.x)} : f32x2{row_bc<0>(sp.x), row_bc<0>(sp.y)};    // (s3, s4)
    const f32x2 shi = TREE ? f32x2{row_bc<8>(sp.x), row_bc<12>(sp.x)} : f32x2{row_bc<1>(sp.x), row_bc<1>(sp.y)};  // (s5, s6)
    // record for the pose / cost waves: the state BEFORE the update (the ring slot is free: see the end of the step);
    // then the publication -- which also says: this wave is done with the control record of step t
    if constexpr (TREE) asm volatile("ds_write_b32 %0, %1" ::"v"(a_rec0 + (uint32_t)slot * kRecStride), "v"(sp.x) : "memory");
    else asm volatile("ds_write_b64 %0, %1" ::"v"(a_rec0 + (uint32_t)slot * kRecStride), "v"(sp) : "memory");
    lds_publish(a_myseq, t + 1);
    // layer 0: [s3, s4, s5, s6, u0, u1]
    f32x2 z = {0.0f, 0.0f};
    z = __builtin_elementwise_fma(W.w1[0], f32x2{slo.x, slo.x}, z);
    z = __builtin_elementwise_fma(W.w1[1], f32x2{slo.y, slo.y}, z);
    z = __builtin_elementwise_fma(W.w1[2], f32x2{shi.x, shi.x}, z);
    z = __builtin_elementwise_fma(W.w1[3], f32x2{shi.y, shi.y}, z);
    z = __builtin_elementwise_fma(W.w1[4], f32x2{u.x, u.x}, z);
    z = __builtin_elementwise_fma(W.w1[5], f32x2{u.y, u.y}, z);
    // Requested now, used at the end of the step: the control wave's count, then this rollout's controls of step t+1 as
    // ONE 8-B read into the pair the packed multiply-adds of layer 0 take them from (valid if the count read before them
    // is >= t+2).  In FRONT of layer 1: the chains below have no LDS wait of their own to hide these reads behind, and a
    // read that is still in flight when a chain starts stalls it (the packed multiply-adds formally read the odd halves of
    // the move registers, which is where the register allocator puts pending results: rollout 55.0 -> 52.4 us with the
    // reads moved here).
    const int sn = ((t + 1) & (kGRing - 1)) * kSlotF2;
    const int cp_v = *p_pub;
    un = p_u[sn];
    const f32x2 a0 = tanh_bias2(z, W.b1s);
    const f32x2 a1 = tanh_bias2(row_dot_bc(W.w2, a0), W.b2s);
    // Step t+1 may start when the control wave has published it (it runs ahead).  That also says that the ring slot of
    // the state record of step t+1 is free -- it held step t+1 - kGRing, consumed once cost_done >= t+2 - kGRing: the
    // control wave publishes a chunk that ends with step tm >= t+1 only after it has seen cost_done >= tm+1 - kGRing
    // (group_control_wave: need_c; the control record of a step shares the slot index of its state record), so this wave
    // does not look at the cost wave's word itself.
    // (A shorter leash for the riders -- waiting when the cost wave is more than 2 / 3 / 5 steps behind instead of a full
    // ring -- was measured: 124 / 72.6 / 58.0 us against 55.2 us; the riders need the slack.)
    // The scalar side of the test in front of the output layer's chain, the (cold) wait behind it.
    const int want = t + 2;
    const int cp_e = __builtin_amdgcn_readfirstlane(cp_v);
    asm volatile("" : "+v"(un));  // the wait for the two reads sits HERE (long arrived), not behind the next step's LDS stores
    if constexpr (TREE) {
      const float d = row_out_tree(W.w3, a1) + W.b3.x;
      sp.x = fmaf(d, a.dt, sp.x);  // incrementState, neural_net_model.cu:334-344
      asm volatile("" : "+v"(sp.x));
    } else {
      const f32x2 d = row_dot_bc(W.w3, a1) + W.b3;
      sp = __builtin_elementwise_fma(d, f32x2{a.dt, a.dt}, sp);  // incrementState, neural_net_model.cu:334-344
      asm volatile("" : "+v"(sp));  // the chain stays here (otherwise it is sunk below the wait, away from its moves)
    }
    if (__builtin_expect(cp_e < want, 0)) {
      int cp = cp_e;
      while (cp < want && --budget > 0) {
        cp = __builtin_amdgcn_readfirstlane(*p_pub);
        un = p_u[sn];
      }
      asm volatile("" : "+v"(un));  // (its wait too: otherwise the merge of the two paths puts one behind the next step's stores)
    }
  }
  if (w == 0) RSTAMP(4);  // T loop done
  {  // the record of step T-1
    const int t = T - 1;
    if constexpr (TREE) asm volatile("ds_write_b32 %0, %1" ::"v"(a_rec0 + (uint32_t)(t & (kGRing - 1)) * kRecStride), "v"(sp.x) : "memory");
    else asm volatile("ds_write_b64 %0, %1" ::"v"(a_rec0 + (uint32_t)(t & (kGRing - 1)) * kRecStride), "v"(sp) : "memory");
    lds_publish(a_myseq, t + 1);
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[w]));
}

// one group (workgroup): the four dynamics waves and the four riders
template <int H, bool AFFINE, bool CTRL, bool TREE, bool GATED = false>
__device__ __forceinline__ void row_group(const RolloutArgs &a, RowShared<H> &sh)
{
  using SH = RowShared<H>;
  using R = GroupRoles<SH>;
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (role == 0) RSTAMP(0);  // first instruction
  MrgHalf g0{0, 0, 0};
  if (role == R::kRng) g0 = group_rng_load<SH>(a);  // in front of the barrier: the head of the launch's critical path
  if (role == 0) {  // sequence words start at 0; the only barrier
#pragma unroll
    for (int w = 0; w < 4; w++) sh.xseq[w][lane] = 0;
    sh.cost_done[lane] = 0;
    sh.ctl_pub[lane] = 0;
    sh.pose_pub[lane] = 0;
    sh.rng_pub[lane] = 0;
    sh.fail[lane & 3] = 0;
    sh.fin[lane & 7] = 0;
    sh.gate_open[lane & 7] = 0;
  }
  __syncthreads();
  if (role == 0) RSTAMP(1);  // behind the barrier
#ifdef MPPI_ROW_RIDER_PRIO
  if (role >= 4) __builtin_amdgcn_s_setprio(MPPI_ROW_RIDER_PRIO);
#endif
  if (role < 4) row_dynamics<H, TREE, GATED>(a, sh, role);
  else if (role == R::kCost) { group_cost_wave4<SH, CTRL>(a, sh); RSTAMP(5); }  // costs stored
  else if (role == R::kCtl) { group_control_wave(a, sh, GATED ? lds_addr(&sh.gate_open[0]) : 0u); RSTAMP(6); }
  else if (role == R::kPose) {
    if constexpr (GATED) {
      const int shut = group_gate_wait(a, sh);
      const volatile float *gs = sh.gstate;
      const float x0 = gs[0], y0 = gs[1], yaw0 = gs[2];
      group_pose_wave4<SH, AFFINE>(a, sh, x0, y0, yaw0, shut);
    } else {
      group_pose_wave4<SH, AFFINE>(a, sh);
    }
    RSTAMP(7);
  }
  else { group_rng_wave<SH, true>(a, sh, g0); RSTAMP(8); }
}

template <int H, bool AFFINE, bool CTRL, bool TREE>
__global__ __launch_bounds__(512) void rollout_row_kernel(const RolloutArgs a)
{
  __shared__ __attribute__((aligned(16))) RowShared<H> sh;
  row_group<H, AFFINE, CTRL, TREE>(a, sh);
}
// the same kernel enqueued one solve ahead (a.gate != nullptr): see row_gate_wait
template <int H, bool AFFINE, bool CTRL, bool TREE>
__global__ __launch_bounds__(512) void rollout_row_gated_kernel(const RolloutArgs a)
{
  __shared__ __attribute__((aligned(16))) RowShared<H> sh;
  row_group<H, AFFINE, CTRL, TREE, true>(a, sh);
}

// several instances in one launch (mppi_compute_control_batch): grid (groups of the largest instance, instances) -- workgroup
// (x, y) runs group x of instance y, whose argument block sits at a position the workgroup knows from its own index
// (MPPI_BATCH_DISPATCH, mppi_device.hpp: one branch per instance, so that the block is read as the single-instance kernel
// reads its arguments -- with a run-time index the block went through scratch: 44.8 us beside 33.8 us alone)
template <int H, bool AFFINE, bool CTRL, bool TREE, int NB>
__global__ __launch_bounds__(512) void rollout_row_batch_kernel(const QuadBatchArgsT<NB> b)
{
  __shared__ __attribute__((aligned(16))) RowShared<H> sh;
#define MPPI_ROW_BODY(A)                                                                                   \
  do {                                                                                                     \
    if ((int)blockIdx.x >= (A).K / kRolloutsPerWave) return; /* a smaller instance than the largest */     \
    row_group<H, AFFINE, CTRL, TREE>((A), sh);                                                             \
  } while (0)
  MPPI_BATCH_DISPATCH(NB, b, MPPI_ROW_BODY);
#undef MPPI_ROW_BODY
}

bool row_variant_supported(int hidden, int n_hidden) { return hidden == 32 && n_hidden == 2; }
int row_pack_floats() { return kRowPackEntries * 16 * 4; }

// The four (AFFINE, CTRL) instances of a kernel template, tree or exact
#define MPPI_ROW_DISPATCH(LAUNCH, KERN, TREE, ...)                                                            \
  do {                                                                                                        \
    if (affine && !ctrl) LAUNCH((KERN<32, true, false, TREE>), __VA_ARGS__);                                  \
    else if (affine && ctrl) LAUNCH((KERN<32, true, true, TREE>), __VA_ARGS__);                               \
    else if (!affine && !ctrl) LAUNCH((KERN<32, false, false, TREE>), __VA_ARGS__);                           \
    else LAUNCH((KERN<32, false, true, TREE>), __VA_ARGS__);                                                  \
  } while (0)
#define MPPI_ROW_BATCH_DISPATCH(TREE, NB, ...)                                                                                  \
  do {                                                                                                                          \
    if (affine && !ctrl) hipLaunchKernelGGL((rollout_row_batch_kernel<32, true, false, TREE, NB>), __VA_ARGS__);                \
    else if (affine && ctrl) hipLaunchKernelGGL((rollout_row_batch_kernel<32, true, true, TREE, NB>), __VA_ARGS__);             \
    else if (!affine && !ctrl) hipLaunchKernelGGL((rollout_row_batch_kernel<32, false, false, TREE, NB>), __VA_ARGS__);         \
    else hipLaunchKernelGGL((rollout_row_batch_kernel<32, false, true, TREE, NB>), __VA_ARGS__);                                \
  } while (0)

hipError_t launch_rollout_row_batch(const QuadBatchArgs &b, bool tree, hipStream_t stream)
{
  if (b.n < 1 || b.n > kMaxBatch) return hipErrorInvalidValue;
  bool affine = true, ctrl = false;  // the general forms are exact supersets (rollout_mfma.hip)
  int gmax = 0;
  for (int i = 0; i < b.n; i++) {
    affine = affine && b.inst[i].cost.affine != 0;
    ctrl = ctrl || b.inst[i].cost.need_control_cost != 0;
    gmax = b.inst[i].K / kRolloutsPerWave > gmax ? b.inst[i].K / kRolloutsPerWave : gmax;
  }
  const dim3 grid(gmax, b.n), block(512);
  if (b.n <= 2) {  // the two controllers of a tick: half the argument segment
    const QuadBatchArgsT<2> b2 = batch_args_prefix<2>(b);
    if (tree) MPPI_ROW_BATCH_DISPATCH(true, 2, grid, block, 0, stream, b2);
    else MPPI_ROW_BATCH_DISPATCH(false, 2, grid, block, 0, stream, b2);
  } else {
    if (tree) MPPI_ROW_BATCH_DISPATCH(true, 4, grid, block, 0, stream, b);
    else MPPI_ROW_BATCH_DISPATCH(false, 4, grid, block, 0, stream, b);
  }
  return hipGetLastError();
}

hipError_t launch_rollout_row(int hidden, int n_hidden, const RolloutArgs &a, bool tree, hipStream_t stream)
{
  if (!row_variant_supported(hidden, n_hidden) || a.K % kRolloutsPerWave != 0) return hipErrorInvalidValue;
  const bool affine = a.cost.affine != 0, ctrl = a.cost.need_control_cost != 0;
  const dim3 grid(a.K / kRolloutsPerWave), block(512);
  if (a.gate != nullptr) {
    if (tree) MPPI_ROW_DISPATCH(MPPI_LAUNCH_ROLLOUT, rollout_row_gated_kernel, true, grid, block, 0, stream, a);
    else MPPI_ROW_DISPATCH(MPPI_LAUNCH_ROLLOUT, rollout_row_gated_kernel, false, grid, block, 0, stream, a);
    return hipGetLastError();
  }
  if (tree) MPPI_ROW_DISPATCH(MPPI_LAUNCH_ROLLOUT, rollout_row_kernel, true, grid, block, 0, stream, a);
  else MPPI_ROW_DISPATCH(MPPI_LAUNCH_ROLLOUT, rollout_row_kernel, false, grid, block, 0, stream, a);
  return hipGetLastError();
}
#undef MPPI_ROW_DISPATCH
#undef MPPI_ROW_BATCH_DISPATCH

}  // namespace mppi

#ifdef MPPI_ROW_STAMPS
extern "C" int mppi_debug_read_row_stamps(unsigned long long *out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mppi::g_row_stamps), sizeof(unsigned long long) * 16);
}
#endif
