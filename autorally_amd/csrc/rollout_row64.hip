// rollout_row64.hip -- rolloutKernel (PI/mppi_controller.cu:72-184) for gfx950, the LATENCY form of 64-wide nets
// (6-64-64-4 and the reference's newest shipped model 6-64-64-64-64-4, params/models/README.md:22) while every group of
// rollouts has a CU of its own: the idea of rollout_row.hip carried over.
//
//   * a rollout is 32 lanes = two 16-lane DPP rows; lane g of the rollout owns neurons 2g, 2g+1 of every hidden layer as a
//     packed pair; a dynamics wave carries two rollouts;
//   * a hidden layer is, per lane, the k-ascending fmaf chain of neural_net_model.cu:379-394 (bias afterwards, the same
//     tanh_bias2): one v_pk_fma_f32 per k with the activation broadcast to both halves.  `row_newbcast` reaches the 16
//     lanes of a row only, so after a layer's tanh every lane fetches the pair of the same lane of the OTHER row of its
//     rollout with one v_permlane16_swap_b32 per component (gfx950): L = the lower row's pair, U = the upper row's, in
//     every lane; activation k then comes from lane (k >> 1) & 15 of L (k < 32) or U by one v_mov_b32_dpp, in the shadow
//     of the multiply-add of k - 1.  No LDS round trip, no partner wave, no poll on the recurrence (the oct form pays
//     ~330 cycles of hand-over per layer on top of the 512 of its chain);
//   * 64 packed weight pairs per lane and 64 x 64 layer are 128 VGPRs: the layers' weights stay in LDS instead --
//     image wl[layer][k][g] = (W[2g][k], W[2g+1][k]), one ds_read_b64 per k requested kPF k-steps ahead; both rollouts of
//     a wave read the same 256 B (2 LDS cycles per wave-instruction, MI355X_MICROARCH.md: LDS).  tools/ub/row64_ub.hip
//     measures the layer alone, from registers and from LDS, with one and two such waves per SIMD;
//   * the OUTPUT layer as in the row-tree form (rollout_row.hip: row_out_tree): lane g multiplies its own two activations
//     into the four outputs and the 32 partials of an output are summed by a butterfly that halves the live values per
//     level -- v_permlane16_swap_b32 (the other row: lower rows keep outputs {0,1}, upper rows {2,3}), row_ror:8 (bit 3 of
//     the lane picks one of the pair), row_half_mirror, two quad_perm -- 2 swaps, one packed add and 4 DPP adds.  NOT the
//     reference's summation order: like "row_tree" the form is checked bit for bit against the test oracle's mode 2 and
//     against the nominal oracle at the north-star tolerance (tests/test_row64_gpu.py).  Lane g ends with output
//     o = 2 (g >> 4) + ((g >> 3) & 1) = state component s[3 + o];
//   * groups of R = 8 rollouts (4 dynamics waves + the 4 riders of group_roles.hpp, one of each per SIMD) while
//     K <= 8 x #CUs -- the reference's K = 1920 --, else R = 16 (8 dynamics waves, two per SIMD, + 4 riders).
#include "group_roles.hpp"
#include "mppi_kernels.hpp"

namespace mppi {

constexpr int kH64 = 64;
constexpr int kPF = 8;  // k-steps a weight pair is requested ahead of its use

template <int NHID, int R>
struct Row64Shared {
  static constexpr int NW = R / 2;        // dynamics waves per group, two rollouts each
  static constexpr int NSW = 1;           // xseq[w] = steps published by dynamics wave w
  static constexpr int kR = R;            // rollouts per group
  static constexpr bool kRecByAll = true; // every dynamics wave writes the state records of its own rollouts
  f32x2 wl[NHID - 1][kH64][32];           // the 64 x 64 layers: [layer][k][lane of the rollout]
  int xseq[NW][64];
  float rec[kGRing][kRolloutsPerWave][4];   // s3..s6 before the update of step t
  int cost_done[64];
  float ctl_b1[kGRing][64];
  float ctl_rec[kGRing][kRolloutsPerWave][4];
  int ctl_pub[64];
  float tex[kGRing][kRolloutsPerWave][2];
  int pose_pub[64];
  float eps[kGRing][kRolloutsPerWave][2];
  int rng_pub[64];
  int fail[4];
  int fin[16];
  float dump[NW][64 * kGRing];  // where the lanes that do not hold a record word put their copy (never read), per ring slot
};

// register part of the image (pack_row64_weights, mppi_abi.hip): 16-B entry i of lane g at float4 index i * 32 + g.
//   0..2            w1: (W1[2g][2i], W1[2g+1][2i], W1[2g][2i+1], W1[2g+1][2i+1])
//   3 .. 3+NB-1     hidden biases x kTanhScale, two layers per entry: (b_l[2g], b_l[2g+1], b_{l+1}[2g], b_{l+1}[2g+1])
//   3+NB, 4+NB      output layer, lane order (see row64_out_tree): (Q for a.x, Q for a.y), (P for a.x, P for a.y)
//   5+NB            (b_out[o], -, -, -)
// followed by the LDS part: (NHID - 1) x 64 x 32 pairs.
template <int NHID>
struct Row64Regs {
  f32x2 w1[kNetIn];
  f32x2 bs[NHID];
  f32x2 q[2], p[2];
  float bo;
};
template <int NHID>
constexpr int row64_reg_entries() { return 3 + (NHID + 1) / 2 + 3; }

template <int NHID>
__device__ __forceinline__ void row64_load(const float *pack, int g, Row64Regs<NHID> &W)
{
  constexpr int NB = (NHID + 1) / 2;
  const float4 *pk = reinterpret_cast<const float4 *>(pack) + g;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float4 v = pk[i * 32];
    W.w1[2 * i] = f32x2{v.x, v.y};
    W.w1[2 * i + 1] = f32x2{v.z, v.w};
  }
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const float4 v = pk[(3 + i) * 32];
    W.bs[2 * i] = f32x2{v.x, v.y};
    if (2 * i + 1 < NHID) W.bs[2 * i + 1] = f32x2{v.z, v.w};
  }
  const float4 u = pk[(3 + NB) * 32], v = pk[(4 + NB) * 32], c = pk[(5 + NB) * 32];
  W.q[0] = f32x2{u.x, u.y};
  W.q[1] = f32x2{u.z, u.w};
  W.p[0] = f32x2{v.x, v.y};
  W.p[1] = f32x2{v.z, v.w};
  W.bo = c.x;
}

template <int Q>
__device__ __forceinline__ float r64_bc(float a)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x150 + Q, 0xF, 0xF, false));  // row_newbcast:Q
}
// (lower row's value, upper row's value) of the same lane position, in every lane of a rollout: v_permlane16_swap_b32 swaps
// the odd rows of its first operand with the even rows of its second -- on two copies of `a` that leaves the lower rows'
// values everywhere in the first and the upper rows' in the second
__device__ __forceinline__ void r64_rows(float a, float &lo, float &up)
{
  auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  lo = __uint_as_float(x[0]);
  up = __uint_as_float(x[1]);
}
// the activation PAIR (a[2P], a[2P+1]) of this lane's rollout: one v_mov_b64_dpp row_newbcast (rollout_row.hip: row_bc2) of the
// lower rows' pair (P < 16) or the upper rows' (P >= 16)
template <int P>
__device__ __forceinline__ f32x2 r64_pair(f32x2 L, f32x2 U)
{
  const f32x2 src = (P < 16) ? L : U;
  return __builtin_bit_cast(f32x2, (long long)__builtin_amdgcn_mov_dpp(__builtin_bit_cast(long long, src), 0x150 + (P & 15), 0xF, 0xF, false));
}
typedef const volatile f32x2 __attribute__((address_space(3))) *r64_lds_p;
// two links of the chain, written as the schedule they have to be (rollout_row.hip: row_dot_step): the weight requests kPF
// links ahead, the multiply-add of 2P, the move for P + 1 in its shadow, the multiply-add of 2P + 1
template <int P>
__device__ __forceinline__ void r64_step(f32x2 &z, f32x2 &b, f32x2 *ring, r64_lds_p base, f32x2 L, f32x2 U)
{
  constexpr int K = 2 * P;
  const f32x2 w0 = ring[K % kPF];
  if constexpr (K + kPF < kH64) ring[K % kPF] = base[(K + kPF) * 32];
  z = __builtin_elementwise_fma(w0, f32x2{b.x, b.x}, z);
  __builtin_amdgcn_sched_barrier(0);
  const f32x2 bn = r64_pair<(P + 1 < kH64 / 2 ? P + 1 : kH64 / 2 - 1)>(L, U);
  const f32x2 w1 = ring[(K + 1) % kPF];
  if constexpr (K + 1 + kPF < kH64) ring[(K + 1) % kPF] = base[(K + 1 + kPF) * 32];
  __builtin_amdgcn_sched_barrier(0);
  z = __builtin_elementwise_fma(w1, f32x2{b.y, b.y}, z);
  __builtin_amdgcn_sched_barrier(0);
  b = bn;
}
template <int P0>
__device__ __forceinline__ void r64_steps16(f32x2 &z, f32x2 &b, f32x2 *ring, r64_lds_p base, f32x2 L, f32x2 U)
{
#define S4(P) r64_step<P>(z, b, ring, base, L, U); r64_step<P + 1>(z, b, ring, base, L, U); r64_step<P + 2>(z, b, ring, base, L, U); r64_step<P + 3>(z, b, ring, base, L, U);
  S4(P0) S4(P0 + 4)
#undef S4
}
// z = sum_k W[.][k] a[k] over the 64 activations of the rollout, k ascending; `base` = this lane's column of the layer's image
__device__ __forceinline__ f32x2 r64_layer(r64_lds_p base, f32x2 a)
{
  f32x2 ring[kPF];
#pragma unroll
  for (int k = 0; k < kPF; k++) ring[k] = base[k * 32];
  float lx, ux, ly, uy;
  r64_rows(a.x, lx, ux);
  r64_rows(a.y, ly, uy);
  const f32x2 L = {lx, ly}, U = {ux, uy};
  f32x2 z = {0.0f, 0.0f};
  f32x2 b = r64_pair<0>(L, U);
  __builtin_amdgcn_sched_barrier(0);
  r64_steps16<0>(z, b, ring, base, L, U);   // (16 links each)
  r64_steps16<8>(z, b, ring, base, L, U);
  r64_steps16<16>(z, b, ring, base, L, U);
  r64_steps16<24>(z, b, ring, base, L, U);
  return z;
}

template <int CTRL>
__device__ __forceinline__ float r64_dpp_add(float acc, float src)
{
  return acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xF, 0xF, true));
}
// The output layer.  Q = this lane's partials of outputs {0, 1}, P = of {2, 3} (both from its own two activations); the swap
// hands the lower rows the upper rows' Q and the upper rows the lower rows' P, so Q + P is then the sum over the lane pair
// (g, g ^ 16) of outputs {0, 1} in the lower rows and {2, 3} in the upper rows.  Inside a pair the lane's KEPT output comes
// first (bit 3 of g picks it: the order is wired into the weights), so the row_ror:8 level needs no select either.
template <int NHID>
__device__ __forceinline__ float row64_out_tree(const Row64Regs<NHID> &W, f32x2 a)
{
  const f32x2 ax = {a.x, a.x}, ay = {a.y, a.y};
  f32x2 q = __builtin_elementwise_fma(W.q[1], ay, W.q[0] * ax);
  f32x2 p = __builtin_elementwise_fma(W.p[1], ay, W.p[0] * ax);
  {
    auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(q.x), __float_as_uint(p.x), false, false);
    auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(q.y), __float_as_uint(p.y), false, false);
    q = f32x2{__uint_as_float(x[0]), __uint_as_float(y[0])};
    p = f32x2{__uint_as_float(x[1]), __uint_as_float(y[1])};
  }
  const f32x2 r = q + p;
  float v = r64_dpp_add<0x128>(r.x, r.y);  // row_ror:8
  v = r64_dpp_add<0x141>(v, v);            // row_half_mirror: lane g <- lane g ^ 7
  v = r64_dpp_add<0xB1>(v, v);             // quad_perm [1,0,3,2]
  v = r64_dpp_add<0x4E>(v, v);             // quad_perm [2,3,0,1]
  return v;
}

template <int NHID, int R>
__device__ __forceinline__ void row64_dynamics(const RolloutArgs &a, Row64Shared<NHID, R> &sh, const int w)
{
  const int lane = threadIdx.x & 63;
  const int g = lane & 31;
  const int jr = 2 * w + (lane >> 5);  // rollout of the group
  const int o = 2 * (g >> 4) + ((g >> 3) & 1);  // the state component this lane carries: s[3 + o]
  const int T = a.T;
  Row64Regs<NHID> W;
  row64_load<NHID>(a.wpack, g, W);
#pragma unroll
  for (int k = 0; k < kNetIn; k++) asm volatile("" : "+v"(W.w1[k]));  // pinned: the waits for the loads sit here

  const uint32_t a_myseq = lds_addr(&sh.xseq[w][lane]);
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  const lds_int_p p_pub = (lds_int_p)&sh.ctl_pub[0];
  const r64_lds_p p_u = (r64_lds_p)&sh.ctl_rec[0][jr][0];  // clamped (u0, u1) of this lane's rollout, ring slot 0
  constexpr int kSlotF2 = kRolloutsPerWave * 2;            // f32x2 per ring slot of ctl_rec
  // the state record of a step: lanes g = 0, 8, 16, 24 hold s3, s4, s5, s6; every lane stores (the others into a dump row
  // nobody reads: no exec masking on the recurrence); both move along with the ring slot
  const uint32_t a_rec0 = ((g & 7) == 0) ? lds_addr(&sh.rec[0][jr][o]) : lds_addr(&sh.dump[w][lane]);
  constexpr uint32_t kRecStride = sizeof(float) * kRolloutsPerWave * 4;
  static_assert(kRecStride == sizeof(float) * 64, "dump rows move along with the record's ring slot");
  r64_lds_p wbase[NHID - 1];
#pragma unroll
  for (int l = 0; l < NHID - 1; l++) wbase[l] = (r64_lds_p)&sh.wl[l][0][g];

  float sv = a.state[3 + o];
  int budget = spin_budget_init(a.spin_budget, T, a.fault_wave == w + 1);
  while (__builtin_amdgcn_readfirstlane(*p_pub) < 1 && --budget > 0) __builtin_amdgcn_s_sleep(1);
  f32x2 un = p_u[0];
  asm volatile("" : "+v"(un));

  // Steps 0 .. T-2 in full; of step T-1 only the state record goes out (its update feeds nothing: the cost is the running
  // mean over the states BEFORE the updates of steps 1..T-1, mppi_controller.cu:160-177)
  for (int t = 0; t < T - 1; t++) {
    const int slot = t & (kGRing - 1);
    const f32x2 u = un;
    float lo, up;
    r64_rows(sv, lo, up);
    const f32x2 s34 = f32x2{r64_bc<0>(lo), r64_bc<8>(lo)}, s56 = f32x2{r64_bc<0>(up), r64_bc<8>(up)};
    asm volatile("ds_write_b32 %0, %1" ::"v"(a_rec0 + (uint32_t)slot * kRecStride), "v"(sv) : "memory");
    lds_publish(a_myseq, t + 1);  // the record is out; also: this wave is done with the control record of step t
    // layer 0: [s3, s4, s5, s6, u0, u1]
    f32x2 z = {0.0f, 0.0f};
    z = __builtin_elementwise_fma(W.w1[0], f32x2{s34.x, s34.x}, z);
    z = __builtin_elementwise_fma(W.w1[1], f32x2{s34.y, s34.y}, z);
    z = __builtin_elementwise_fma(W.w1[2], f32x2{s56.x, s56.x}, z);
    z = __builtin_elementwise_fma(W.w1[3], f32x2{s56.y, s56.y}, z);
    z = __builtin_elementwise_fma(W.w1[4], f32x2{u.x, u.x}, z);
    z = __builtin_elementwise_fma(W.w1[5], f32x2{u.y, u.y}, z);
    // requested now, used at the end of the step (rollout_row.hip): the control wave's count and this rollout's controls
    // of step t+1 (valid if the count read before them is >= t+2)
    const int sn = ((t + 1) & (kGRing - 1)) * kSlotF2;
    const int cp_v = *p_pub;
    un = p_u[sn];
    f32x2 act = tanh_bias2(z, W.bs[0]);
#pragma unroll
    for (int l = 1; l < NHID; l++) act = tanh_bias2(r64_layer(wbase[l - 1], act), W.bs[l]);
    const int want = t + 2;
    const int cp_e = __builtin_amdgcn_readfirstlane(cp_v);
    asm volatile("" : "+v"(un));
    {
      const float d = row64_out_tree<NHID>(W, act) + W.bo;
      sv = fmaf(d, a.dt, sv);  // incrementState, neural_net_model.cu:334-344
      asm volatile("" : "+v"(sv));
    }
    // Step t+1 may start when the control wave has published it; that also says that the ring slot of the state record of
    // step t+1 is free (group_control_wave: need_c -- rollout_row.hip has the argument)
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

template <int NHID, int R, bool AFFINE, bool CTRL>
__global__ __launch_bounds__(64 * (R / 2 + 4)) void rollout_row64_kernel(const RolloutArgs a)
{
  using SH = Row64Shared<NHID, R>;
  using RO = GroupRoles<SH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SH &sh = *reinterpret_cast<SH *>(smem_raw);
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  MrgHalf g0{0, 0, 0};
  if (role == RO::kRng) g0 = group_rng_load<SH>(a);  // in front of the barrier: the head of the launch's critical path
  {  // the 64 x 64 layers into LDS: the image is in LDS order, 16 B per thread and pass
    const float4 *src = reinterpret_cast<const float4 *>(a.wpack) + row64_reg_entries<NHID>() * 32;
    float4 *dst = reinterpret_cast<float4 *>(&sh.wl[0][0][0]);
    constexpr int n4 = (NHID - 1) * kH64 * 32 / 2;
    for (int i = threadIdx.x; i < n4; i += 64 * (R / 2 + 4)) dst[i] = src[i];
  }
  if (role == 0) {  // sequence words start at 0
#pragma unroll
    for (int w = 0; w < SH::NW; w++) sh.xseq[w][lane] = 0;
    sh.cost_done[lane] = 0;
    sh.ctl_pub[lane] = 0;
    sh.pose_pub[lane] = 0;
    sh.rng_pub[lane] = 0;
    sh.fail[lane & 3] = 0;
    sh.fin[lane & 15] = 0;
  }
  __syncthreads();  // the only barrier
  if (role < SH::NW) row64_dynamics<NHID, R>(a, sh, role);
  else if (role == RO::kCost) group_cost_wave4<SH, CTRL>(a, sh);
  else if (role == RO::kCtl) group_control_wave(a, sh);
  else if (role == RO::kPose) group_pose_wave4<SH, AFFINE>(a, sh);
  else group_rng_wave<SH, true>(a, sh, g0);
}

bool row64_variant_supported(int hidden, int n_hidden) { return hidden == 64 && (n_hidden == 2 || n_hidden == 4); }
int row64_pack_floats(int n_hidden)
{
  return (n_hidden == 2 ? row64_reg_entries<2>() : row64_reg_entries<4>()) * 32 * 4 + (n_hidden - 1) * kH64 * 64;
}

template <int NHID, int R>
static hipError_t launch_row64(const RolloutArgs &a, hipStream_t stream)
{
  const bool affine = a.cost.affine != 0, ctrl = a.cost.need_control_cost != 0;
  const dim3 grid(a.K / R), block(64 * (R / 2 + 4));
  const size_t lds = sizeof(Row64Shared<NHID, R>);
#define MPPI_R64(AF, CT)                                                                                              \
  do {                                                                                                                \
    static bool attr_set[64] = {}; /* more dynamic LDS than the default limit: once per kernel instance and device */ \
    int dev = 0;                                                                                                      \
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;                       \
    if (!attr_set[dev]) {                                                                                             \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_row64_kernel<NHID, R, AF, CT>),      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
      if (e != hipSuccess) return e;                                                                                  \
      attr_set[dev] = true;                                                                                           \
    }                                                                                                                 \
    MPPI_LAUNCH_ROLLOUT((rollout_row64_kernel<NHID, R, AF, CT>), grid, block, lds, stream, a);                        \
  } while (0)
  if (affine && !ctrl) MPPI_R64(true, false);
  else if (affine && ctrl) MPPI_R64(true, true);
  else if (!affine && !ctrl) MPPI_R64(false, false);
  else MPPI_R64(false, true);
#undef MPPI_R64
  return hipGetLastError();
}

// r: rollouts per group, 8 (K a multiple of 8) or 16
hipError_t launch_rollout_row64(int hidden, int n_hidden, const RolloutArgs &a, int r, hipStream_t stream)
{
  if (!row64_variant_supported(hidden, n_hidden) || (r != 8 && r != 16) || a.K % r != 0) return hipErrorInvalidValue;
  if (r != 16) return hipErrorInvalidValue;  // (round 4 also built groups of 8 rollouts: never faster than the oct / m44 forms, removed)
  return n_hidden == 2 ? launch_row64<2, 16>(a, stream) : launch_row64<4, 16>(a, stream);
}

}  // namespace mppi
