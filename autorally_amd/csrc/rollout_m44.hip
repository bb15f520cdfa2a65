// rollout_m44.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) for gfx950, the latency form of 64-WIDE nets on
// v_mfma_f32_4x4x1 with A-matrix broadcast (6-64-64-4 and the reference's newest shipped model 6-64-64-64-64-4,
// params/models/README.md:22) while every group of 16 rollouts has a CU of its own.
//
// v_mfma_f32_4x4x1_16b_f32 multiplies, in each of 16 blocks, a 4 x 1 A by a 1 x 4 B into a 4 x 4 D (+= C); with CBSZ = 4,
// ABID = b' ALL 16 blocks take block b''s four A values (tools/ub/mfma4x4_ub.hip probes the layouts).  So with
//   A = a VGPR whose lanes 4 b' + i hold the activation a_i[k] of rollouts i = 0..3      (rows of D = rollouts)
//   B = a VGPR whose lane n holds W[n][k]                                                  (columns of D = neurons)
// ONE instruction is step k of the k-ascending fmaf chain of neural_net_model.cu:379-394 for 4 rollouts x 64 neurons: the
// broadcast of a_i[k] costs no instruction, and a layer's weights are 64 VGPRs per lane (a lane holds ITS neuron's row, no
// copy per rollout) -- the three 64 x 64 layers of 6-64-64-64-64-4 fit the registers of one wave (192 of 256), which no
// vector-ALU layout does (rollout_row64.hip has to read them from LDS and loses to the oct form for it:
// profiles/r04_c_row64_first.txt).  Measured alone: 960 ticks per 64 x 64 layer + tanh + transpose for the 4 rollouts of a
// wave (profiles/r04_c_mfma4x4_ub.txt; the oct form's layer is ~840 cycles for 16 rollouts but needs four waves and an
// all-to-all hand-over per layer, its step 4 200 cycles at 6-64-64-64-64-4).
//   * a dynamics wave = 4 rollouts; 4 dynamics waves + the 4 riders of group_roles.hpp per 16 rollouts, as in the row form;
//   * D leaves a layer as VGPR r = rollout r, lane n = neuron n; the next layer wants lane-in-quad = rollout, VGPR =
//     neuron-in-quad: a 4 x 4 transpose inside every quad, 4 DPP moves + 12 selects, once per layer;
//   * layer 0 takes the state straight from the state register: row c of the wave (lanes 16 c ..) holds s[3 + c] of rollout
//     lane & 3, so ABID = 4 c picks it -- no broadcast instruction anywhere on the recurrence;
//   * the OUTPUT layer is a reduction over the lanes: in the transposed layout lane 4 b + i holds activations 4 b .. 4 b + 3 of
//     rollout i -- it multiplies them into the four outputs (a product and three fused multiply-adds each, packed) and the 16
//     blocks are summed by a butterfly that halves the live values: v_permlane32_swap (lower half-wave keeps outputs {0,1}),
//     v_permlane16_swap (even rows keep the first of the pair), row_ror:8, row_ror:4.  Row c ends with output c.  NOT the
//     reference's summation order;
//   * the HIDDEN layers: template parameter SPLIT.  SPLIT (the AUTOMATIC form for 64-wide nets up to 8192 rollouts since the
//     end of round 4, variant "m44", name "..._m44_split_tree"): the even-k and the odd-k matrix instructions accumulate
//     into two registers that are added at the end -- two independent chains issue at the pipe's 8 cycles instead of one
//     dependent chain's 12.2.  NOT the reference's order either: the test oracle states it as mode 5.  !SPLIT (variant
//     "m44_chain", name "..._m44_tree", oracle mode 3): one chain per hidden layer, the reference's k-ascending order --
//     a maintainer comparing bits of hidden-layer activations with the CUDA binary wants this one, or "mfma" (the oct form:
//     the reference's order in EVERY layer).
//     Both are checked bit for bit against their oracle mode and against the NOMINAL oracle at the north-star tolerance
//     (tests/test_m44_gpu.py); what the re-association costs at the 1e-4 mark, at the launch defaults:
//     profiles/r05_b_nominal_margin_wd.txt (no more draws beyond 1e-4 than the exact oct form).
#include "group_roles.hpp"
#include "mppi_kernels.hpp"

namespace mppi {

typedef float m44_f4 __attribute__((ext_vector_type(4)));
constexpr int kM44H = 64;

struct M44Shared {
  static constexpr int NW = 4;            // dynamics waves per group, four rollouts each
  static constexpr int NSW = 1;
  static constexpr int kR = 16;
  static constexpr bool kRecByAll = true;
  int xseq[NW][64];
  float rec[kGRing][kRolloutsPerWave][4];
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
  float gstate[8];   // gated launch: the vehicle state the pose wave took from the gate block, then 1 in gate_open[]
  int gate_open[8];
  m44_f4 wo[4][64];             // output-layer weights of lane l (6-64x4-4: they do not fit the registers beside 192 + ...)
  float dump[NW][64 * kGRing];  // where the lanes that hold no record word put their copy (never read), per ring slot
};

// Image (pack_m44_weights, mppi_abi.hip): float4 q of lane l at float4 index q * 64 + l; as floats e = 4 q + c:
//   e 0..5                 W0[l][c]                    (layer 0, B operand of k = c)
//   e 6..7                 0
//   e 8 .. 8+NHID-1        b_layer[l] x kTanhScale     (padded to a multiple of 4)
//   then 64 per hidden layer 1..NHID-1:  W_layer[l][k], k = 0..63
//   then 16:               output layer for lane l = 4 b + i: (W3[0][4b+s], W3[1][4b+s]) s = 0..3, then (W3[2][..], W3[3][..])
//   then 4:                (b_out[l >> 4], 0, 0, 0)
template <int NHID>
constexpr int m44_q_bias() { return 2; }
template <int NHID>
constexpr int m44_q_hidden() { return 2 + (NHID + 3) / 4; }
template <int NHID>
constexpr int m44_q_out() { return m44_q_hidden<NHID>() + (NHID - 1) * 16; }
template <int NHID>
constexpr int m44_q_total() { return m44_q_out<NHID>() + 5; }

template <int Q>
__device__ __forceinline__ float m44_qp(float v)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q, 0xF, 0xF, false));
}
// 4 x 4 transpose inside every quad: in[r] lane 4 b + j  ->  out[s] lane 4 b + i = in[i] lane 4 b + s
__device__ __forceinline__ void m44_transpose(const float (&in)[4], float (&out)[4], bool hi, bool od)
{
  float t[4];
  {  // exchange the off-diagonal 2 x 2 blocks (registers r <-> r ^ 2, lanes ^ 2)
    const float x0 = m44_qp<0x4E>(hi ? in[0] : in[2]);  // quad_perm [2,3,0,1]
    const float x1 = m44_qp<0x4E>(hi ? in[1] : in[3]);
    t[0] = hi ? x0 : in[0];
    t[2] = hi ? in[2] : x0;
    t[1] = hi ? x1 : in[1];
    t[3] = hi ? in[3] : x1;
  }
  {  // inside each 2 x 2 block (registers r <-> r ^ 1, lanes ^ 1)
    const float y0 = m44_qp<0xB1>(od ? t[0] : t[1]);  // quad_perm [1,0,3,2]
    const float y1 = m44_qp<0xB1>(od ? t[2] : t[3]);
    out[0] = od ? y0 : t[0];
    out[1] = od ? t[1] : y0;
    out[2] = od ? y1 : t[2];
    out[3] = od ? t[3] : y1;
  }
}

template <int K>
__device__ __forceinline__ void m44_step(m44_f4 &d, const float (&T)[4], const float *w)
{
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(T[K & 3], w[K], d, 4, K >> 2, 0);
}
template <int K0>
__device__ __forceinline__ void m44_steps16(m44_f4 &d, const float (&T)[4], const float *w)
{
#define S4(K) m44_step<K>(d, T, w); m44_step<K + 1>(d, T, w); m44_step<K + 2>(d, T, w); m44_step<K + 3>(d, T, w);
  S4(K0) S4(K0 + 4) S4(K0 + 8) S4(K0 + 12)
#undef S4
}
__device__ __forceinline__ void m44_tanh(const m44_f4 &d, float bs, float (&act)[4])
{
  const f32x2 a01 = tanh_bias2(f32x2{d[0], d[1]}, f32x2{bs, bs});
  const f32x2 a23 = tanh_bias2(f32x2{d[2], d[3]}, f32x2{bs, bs});
  act[0] = a01.x; act[1] = a01.y; act[2] = a23.x; act[3] = a23.y;
}

template <int CTRL>
__device__ __forceinline__ float m44_dpp_add(float acc, float src)
{
  return acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xF, 0xF, true));
}
// The output layer from the transposed activations T[s] (lane 4 b + i: activation 4 b + s of rollout i).  q[s] / p[s]: this
// lane's weights of outputs {0, 1} / {2, 3} for its activation s.  Q = partials of outputs {0, 1}, P = of {2, 3}: after
// v_permlane32_swap(Q, P) the lower half-wave holds the upper half's Q in P and the upper half the lower half's P in Q, so
// Q + P is the sum over the lane pair (l, l ^ 32) of outputs {0, 1} below and {2, 3} above; the same once more with
// v_permlane16_swap on the two components.  Row c then holds output c, summed over its own four quads by row_ror:8, row_ror:4.
__device__ __forceinline__ float m44_out_tree(const f32x2 (&q)[4], const f32x2 (&p)[4], const float (&T)[4])
{
  f32x2 Q = q[0] * f32x2{T[0], T[0]}, P = p[0] * f32x2{T[0], T[0]};
#pragma unroll
  for (int s = 1; s < 4; s++) {
    Q = __builtin_elementwise_fma(q[s], f32x2{T[s], T[s]}, Q);
    P = __builtin_elementwise_fma(p[s], f32x2{T[s], T[s]}, P);
  }
  {
    auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(Q.x), __float_as_uint(P.x), false, false);
    auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(Q.y), __float_as_uint(P.y), false, false);
    Q = f32x2{__uint_as_float(x[0]), __uint_as_float(y[0])};
    P = f32x2{__uint_as_float(x[1]), __uint_as_float(y[1])};
  }
  const f32x2 r = Q + P;
  auto z = __builtin_amdgcn_permlane16_swap(__float_as_uint(r.x), __float_as_uint(r.y), false, false);
  float v = __uint_as_float(z[0]) + __uint_as_float(z[1]);
  v = m44_dpp_add<0x128>(v, v);  // row_ror:8
  v = m44_dpp_add<0x124>(v, v);  // row_ror:4
  return v;
}

template <int NHID, bool SPLIT, bool GATED = false>
__device__ __forceinline__ void m44_dynamics(const RolloutArgs &a, M44Shared &sh, const int w)
{
  constexpr bool OUT_LDS = (NHID > 2);
  const int lane = threadIdx.x & 63;
  const int i = lane & 3, row = lane >> 4;
  const int jr = 4 * w + i;  // rollout of the group (A layout: lane-in-quad = rollout)
  const bool hi = (lane & 2) != 0, od = (lane & 1) != 0;
  const int T = a.T;
  const float4 *pk = reinterpret_cast<const float4 *>(a.wpack) + lane;
  float w0[8], bsv[4 * ((NHID + 3) / 4)], wh[NHID - 1][kM44H];
  {
    const float4 u = pk[0], v = pk[64];
    w0[0] = u.x; w0[1] = u.y; w0[2] = u.z; w0[3] = u.w; w0[4] = v.x; w0[5] = v.y; w0[6] = v.z; w0[7] = v.w;
#pragma unroll
    for (int q = 0; q < (NHID + 3) / 4; q++) {
      const float4 b = pk[(m44_q_bias<NHID>() + q) * 64];
      bsv[4 * q] = b.x; bsv[4 * q + 1] = b.y; bsv[4 * q + 2] = b.z; bsv[4 * q + 3] = b.w;
    }
#pragma unroll
    for (int l = 0; l < NHID - 1; l++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const float4 x = pk[(m44_q_hidden<NHID>() + 16 * l + q) * 64];
        wh[l][4 * q] = x.x; wh[l][4 * q + 1] = x.y; wh[l][4 * q + 2] = x.z; wh[l][4 * q + 3] = x.w;
      }
  }
  f32x2 oq[4], op[4];
  if constexpr (!OUT_LDS) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const float4 x = pk[(m44_q_out<NHID>() + s) * 64], y = pk[(m44_q_out<NHID>() + 2 + s) * 64];
      oq[2 * s] = f32x2{x.x, x.y}; oq[2 * s + 1] = f32x2{x.z, x.w};
      op[2 * s] = f32x2{y.x, y.y}; op[2 * s + 1] = f32x2{y.z, y.w};
    }
  }
  const float bo = pk[(m44_q_out<NHID>() + 4) * 64].x;
  // pinned: the waits for the weight loads sit here, not at their first use inside the T loop
#pragma unroll
  for (int l = 0; l < NHID - 1; l++)
#pragma unroll
    for (int k = 0; k < kM44H; k++) asm volatile("" : "+v"(wh[l][k]));

  const uint32_t a_myseq = lds_addr(&sh.xseq[w][lane]);
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile f32x2 __attribute__((address_space(3))) *lds_f2_p;
  typedef const volatile m44_f4 __attribute__((address_space(3))) *lds_f4_p;
  const lds_int_p p_pub = (lds_int_p)&sh.ctl_pub[0];
  const lds_f2_p p_u = (lds_f2_p)&sh.ctl_rec[0][jr][0];  // clamped (u0, u1) of rollout lane & 3, ring slot 0
  const lds_f4_p p_wo = (lds_f4_p)&sh.wo[0][lane];
  constexpr int kSlotF2 = kRolloutsPerWave * 2;
  // the state record: quad 0 of row c holds s[3 + c] of rollouts 0..3; every lane stores (the others into a dump row)
  const uint32_t a_rec0 = ((lane & 12) == 0) ? lds_addr(&sh.rec[0][jr][row]) : lds_addr(&sh.dump[w][lane]);
  constexpr uint32_t kRecStride = sizeof(float) * kRolloutsPerWave * 4;
  static_assert(kRecStride == sizeof(float) * 64, "dump rows move along with the record's ring slot");

  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == w + 1);
  float sv;
  if constexpr (GATED) {  // the state arrives through the gate block: the pose wave has put it into LDS (group_gate_wait)
    const uint32_t a_go = lds_addr(&sh.gate_open[0]);
    while (lds_peek(a_go) == 0 && --budget > 0) __builtin_amdgcn_s_sleep(1);
    const volatile float *gs = sh.gstate;
    sv = gs[3 + row];
  } else {
    sv = a.state[3 + row];
  }
  while (__builtin_amdgcn_readfirstlane(*p_pub) < 1 && --budget > 0) __builtin_amdgcn_s_sleep(1);
  f32x2 un = p_u[0];
  asm volatile("" : "+v"(un));

  for (int t = 0; t < T - 1; t++) {
    const int slot = t & (kGRing - 1);
    const f32x2 u = un;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a_rec0 + (uint32_t)slot * kRecStride), "v"(sv) : "memory");
    lds_publish(a_myseq, t + 1);  // the record is out; also: this wave is done with the control record of step t
    // layer 0: [s3, s4, s5, s6, u0, u1] -- row c of the state register is component c: ABID = 4 c
    m44_f4 d = {0.0f, 0.0f, 0.0f, 0.0f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(sv, w0[0], d, 4, 0, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(sv, w0[1], d, 4, 4, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(sv, w0[2], d, 4, 8, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(sv, w0[3], d, 4, 12, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(u.x, w0[4], d, 4, 0, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(u.y, w0[5], d, 4, 0, 0);
    // requested now, used at the end of the step (rollout_row.hip)
    const int sn = ((t + 1) & (kGRing - 1)) * kSlotF2;
    const int cp_v = *p_pub;
    un = p_u[sn];
    float act[4], Tr[4];
    m44_tanh(d, bsv[0], act);
#pragma unroll
    for (int l = 1; l < NHID; l++) {
      m44_transpose(act, Tr, hi, od);
      if constexpr (OUT_LDS) {
        if (l == NHID - 1) {  // the output layer's weights, requested in front of the last chain: they arrive under it
#pragma unroll
          for (int s = 0; s < 2; s++) {
            const m44_f4 x = p_wo[s * 64], y = p_wo[(2 + s) * 64];
            oq[2 * s] = f32x2{x.x, x.y}; oq[2 * s + 1] = f32x2{x.z, x.w};
            op[2 * s] = f32x2{y.x, y.y}; op[2 * s + 1] = f32x2{y.z, y.w};
          }
        }
      }
      if constexpr (SPLIT) {
        // TWO accumulation chains, even / odd k, added at the end: a dependent v_mfma_f32_4x4x1 issues every 12.2 cycles, two
        // independent ones every ~8 -- a 64-input layer 780 -> ~520 cycles (6-64x4-4, K=1920: rollout 148.5 -> 119.0 us).  NOT the
        // reference's order (neural_net_model.cu:379-394 sums k ascending): the form's own oracle mode is 5, and it is held
        // against the nominal oracle at the north-star criteria like the butterfly output layer (tests/test_m44_gpu.py)
        m44_f4 d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = {0.0f, 0.0f, 0.0f, 0.0f};
#define S2(K) m44_step<K>(d0, Tr, wh[l - 1]); m44_step<K + 1>(d1, Tr, wh[l - 1]);
#define S8(K) S2(K) S2(K + 2) S2(K + 4) S2(K + 6)
        S8(0) S8(8) S8(16) S8(24) S8(32) S8(40) S8(48) S8(56)
#undef S8
#undef S2
        d = d0 + d1;
      } else {
        d = m44_f4{0.0f, 0.0f, 0.0f, 0.0f};
        m44_steps16<0>(d, Tr, wh[l - 1]);
        m44_steps16<16>(d, Tr, wh[l - 1]);
        m44_steps16<32>(d, Tr, wh[l - 1]);
        m44_steps16<48>(d, Tr, wh[l - 1]);
      }
      m44_tanh(d, bsv[l], act);
    }
    m44_transpose(act, Tr, hi, od);
    const int want = t + 2;
    const int cp_e = __builtin_amdgcn_readfirstlane(cp_v);
    asm volatile("" : "+v"(un));
    {
      const float dd = m44_out_tree(oq, op, Tr) + bo;
      sv = fmaf(dd, a.dt, sv);  // incrementState, neural_net_model.cu:334-344
      asm volatile("" : "+v"(sv));
    }
    if (__builtin_expect(cp_e < want, 0)) {
      int cp = cp_e;
      while (cp < want && --budget > 0) {
        cp = __builtin_amdgcn_readfirstlane(*p_pub);
        un = p_u[sn];
      }
      asm volatile("" : "+v"(un));
    }
  }
  {  // the record of step T-1
    const int t = T - 1;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a_rec0 + (uint32_t)(t & (kGRing - 1)) * kRecStride), "v"(sv) : "memory");
    lds_publish(a_myseq, t + 1);
  }
  spin_finish(budget, lds_addr(&sh.fail[0]), lds_addr(&sh.fin[w]));
}

// GATED: enqueued one solve ahead (a.gate != nullptr), state and nominal sequence from the gate block: group_gate_wait
template <int NHID, bool AFFINE, bool CTRL, bool SPLIT, bool GATED = false>
__global__ __launch_bounds__(512) void rollout_m44_kernel(const RolloutArgs a)
{
  using SH = M44Shared;
  using RO = GroupRoles<SH>;
  __shared__ __attribute__((aligned(16))) SH sh;
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  MrgHalf g0{0, 0, 0};
  if (role == RO::kRng) g0 = group_rng_load<SH>(a);
  if (role == 0) {
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
  if (role == 1) {  // the output layer's weights into LDS
    const m44_f4 *src = reinterpret_cast<const m44_f4 *>(a.wpack) + m44_q_out<NHID>() * 64 + lane;
#pragma unroll
    for (int s = 0; s < 4; s++) sh.wo[s][lane] = src[s * 64];
  }
  __syncthreads();  // the only barrier
  if (role < 4) m44_dynamics<NHID, SPLIT, GATED>(a, sh, role);
  else if (role == RO::kCost) group_cost_wave4<SH, CTRL>(a, sh);
  else if (role == RO::kCtl) group_control_wave(a, sh, GATED ? lds_addr(&sh.gate_open[0]) : 0u);
  else if (role == RO::kPose) {
    if constexpr (GATED) {
      const int shut = group_gate_wait(a, sh);
      const volatile float *gs = sh.gstate;
      const float x0 = gs[0], y0 = gs[1], yaw0 = gs[2];
      group_pose_wave4<SH, AFFINE>(a, sh, x0, y0, yaw0, shut);
    } else {
      group_pose_wave4<SH, AFFINE>(a, sh);
    }
  }
  else group_rng_wave<SH, true>(a, sh, g0);
}

bool m44_variant_supported(int hidden, int n_hidden) { return hidden == 64 && (n_hidden == 2 || n_hidden == 4); }
int m44_pack_floats(int n_hidden) { return (n_hidden == 2 ? m44_q_total<2>() : m44_q_total<4>()) * 64 * 4; }

template <int NHID, bool SPLIT>
static hipError_t launch_m44(const RolloutArgs &a, hipStream_t stream)
{
  const bool affine = a.cost.affine != 0, ctrl = a.cost.need_control_cost != 0;
  const dim3 grid(a.K / kRolloutsPerWave), block(512);
  if (a.gate != nullptr) {  // the gated form exists for the automatic (split) form only: abi_solve.hip: chain_ok
    if constexpr (SPLIT) {
      if (affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_m44_kernel<NHID, true, false, true, true>), grid, block, 0, stream, a);
      else if (affine && ctrl) MPPI_LAUNCH_ROLLOUT((rollout_m44_kernel<NHID, true, true, true, true>), grid, block, 0, stream, a);
      else if (!affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_m44_kernel<NHID, false, false, true, true>), grid, block, 0, stream, a);
      else MPPI_LAUNCH_ROLLOUT((rollout_m44_kernel<NHID, false, true, true, true>), grid, block, 0, stream, a);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_m44_kernel<NHID, true, false, SPLIT>), grid, block, 0, stream, a);
  else if (affine && ctrl) MPPI_LAUNCH_ROLLOUT((rollout_m44_kernel<NHID, true, true, SPLIT>), grid, block, 0, stream, a);
  else if (!affine && !ctrl) MPPI_LAUNCH_ROLLOUT((rollout_m44_kernel<NHID, false, false, SPLIT>), grid, block, 0, stream, a);
  else MPPI_LAUNCH_ROLLOUT((rollout_m44_kernel<NHID, false, true, SPLIT>), grid, block, 0, stream, a);
  return hipGetLastError();
}

// split: the hidden layers as two accumulation chains (the automatic form); false: one chain, the reference's order ("m44_chain")
hipError_t launch_rollout_m44(int hidden, int n_hidden, const RolloutArgs &a, bool split, hipStream_t stream)
{
  if (!m44_variant_supported(hidden, n_hidden) || a.K % kRolloutsPerWave != 0) return hipErrorInvalidValue;
  if (n_hidden == 2) return split ? launch_m44<2, true>(a, stream) : launch_m44<2, false>(a, stream);
  return split ? launch_m44<4, true>(a, stream) : launch_m44<4, false>(a, stream);
}

}  // namespace mppi
