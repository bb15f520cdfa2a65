// mppi_device.hpp -- argument blocks and device helpers shared by the gfx950 kernels.
//
// All kernels are compiled with -ffp-contract=off: every fused multiply-add is an
// explicit fmaf()/MFMA placed where the reference's nvcc build contracts one
// (SURVEY 8c "fidelity rules"); everything else rounds exactly as written.
// The per-step helpers are written branch-free (selects instead of if/else) so that one
// rollout step is a single basic block and the scheduler can interleave the cost / kinematics
// arithmetic with the MFMA chain of the network.
// Reference citations are relative to /root/reference/autorally_control/,
//   PI/ = include/autorally_control/path_integral/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mppi {

constexpr int kStateDim = 7;
constexpr int kControlDim = 2;
constexpr int kNetIn = 6;   // [roll, u_x, u_y, yaw_mder, steering, throttle]
constexpr int kNetOut = 4;  // d/dt [roll, u_x, u_y, yaw_mder]
constexpr int kRolloutsPerWave = 16;  // N dimension of v_mfma_f32_16x16x4_f32

// Scalars of MPPICosts::CostParams (PI/costs.cuh:67-85) in the form the kernels consume.
struct CostArgs {
  float desired_speed, speed_coeff, track_coeff, max_slip_ang, slip_penalty, track_slop;
  float crash_coeff, steering_coeff, throttle_coeff, boundary_threshold;
  float crash_cost_discounted;  // (float)((1.0 - (double)discount) * (double)crash_coeff), costs.cu:402 (Q6)
  int l1_cost;
  float r_c1[3], r_c2[3], trs[3];
  int affine;     // r_c1.z == 0 && r_c2.z == 0 && trs.z == 1  => w == 1 exactly, u/w == u
  int need_control_cost;  // steering_coeff != 0 || throttle_coeff != 0 || nu not finite/non-zero
  int map_w, map_h;
  const float *map;  // channel 0 plane, [H][W]
};

struct RolloutArgs {
  float state[kStateDim];
  const float *U;   // [T][2]
  float *noise;     // [T][K][2]: in N(0,1), out applied unclamped control (Q3)
  float *costs;     // [K]
  const float *wpack;  // MFMA-ordered weights (see pack_mfma_weights) or packed theta (VALU kernel)
  const double *inv_t; // inv_t[t] = RN(1.0 / t) in double, t = 1..T-1 (running-mean division)
  // in-kernel noise (quad kernel only): per-rollout MRG32k3a states [6][K], read from rng_in and
  // written (advanced by 2T draws) to rng_out; inline_noise == 0 => eps is read from `noise`
  const uint32_t *rng_in;
  uint32_t *rng_out;
  int inline_noise;
  int K, T, opt_delay, k99;
  float nu[2], u_lo[2], u_hi[2], dt;
  int negate_yaw_der;
  // hand-over waits (multi-wavefront kernels): polls a wave may spend on all its waits (0: kSpinBudget),
  // and -- tests only -- the wavefront role (1-based, 0 = none) that starts with that budget exhausted
  // (mppi_debug_inject_handover_fault)
  int spin_budget, fault_wave;
  // gated launch (rollout_row.hip, the chained control ticks of abi_solve.hip): the kernel was enqueued one solve ahead; its
  // vehicle state is NOT `state` above but the 7 floats of the gate block, valid once word 7 of the block equals gate_seq
  // (kGateReplicas copies of 64 B each, written by the host; workgroup b polls copy b % kGateReplicas).  nullptr: not gated.
  const unsigned *gate;
  unsigned gate_seq;
  // the launch's minimum cost for the tail kernel that follows (publish_min_cost below): kMinCostLines keys, the launch's tag.
  // nullptr: not published (the tail takes the minimum itself)
  unsigned long long *min_cost;
  unsigned min_cost_tag;
  CostArgs cost;
};
constexpr int kGateReplicas = 8;
// behind the replicas: the nominal control sequence U[T][2] of the gated solve and the control history hist[4] its tail kernel
// smooths with -- one copy each, written by the host before the gate words (float offsets into the block)
constexpr int kGateUOffset = 16 * kGateReplicas;
inline int gate_hist_offset(int T) { return kGateUOffset + 2 * T; }
inline size_t gate_block_floats(int T) { return (size_t)gate_hist_offset(T) + 4; }
constexpr unsigned kGateCancel = 0x80000000u;  // gate word = gate_seq | kGateCancel: the solve is called off (costs poisoned)

// ---- beta = min_k costs[k] on its way out of the rollout kernel (round 5) ----
// The tail stage's first step is the minimum of all K costs (mppi_controller.cu:630-634: computeNormalizer's baseline).  It is
// exact and order-free, so the waves that write costs[] leave it behind themselves: a wave's minimum (NaN and +inf left out,
// as fminf over the costs leaves them out) goes as ONE 64-bit atomic minimum of the key {~tag, order-preserving bits of the
// cost} to line blockIdx.x % kMinCostLines.  A later launch's tag is larger, its ~tag smaller: keys of this launch beat whatever
// older launches left in a line, so nothing is ever reset; the tail takes the minimum of the kMinCostLines keys and uses it when
// its tag is this launch's -- otherwise (a form that does not publish, every cost NaN or +inf) it reduces the costs itself as
// before.  Same bits either way.  Eight lines: same-line atomics are served one after another (~100 per us); a launch of
// 16 384 rollouts sends 256.  Published by rollout_multi.hip's forms, for solves whose tail is the streaming kernel
// (abi_solve.hip: min_cost_keys -- with the row / m44 forms of K <= 8192 it was measured a loss).
constexpr int kMinCostLines = 8, kMinCostStride = 16;  // keys; 128 B apart
__device__ __forceinline__ unsigned cost_order_bits(float x)
{
  const unsigned b = __float_as_uint(x);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float cost_from_order_bits(unsigned o)
{
  return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// Wave-wide reductions that stay out of the LDS pipeline (ds_bpermute: ~100 cycles a step): two quad_perm steps, then
// row_half_mirror and row_mirror (after the quad steps a quad's lanes are equal, so the mirrored lane holds "the other quad" /
// "the other half"), then the two cross-row steps of gfx950: v_permlane16_swap / v_permlane32_swap on two copies of the
// value leave the even row's (lower half's) value in the first result and the odd row's (upper half's) in the second, in
// every lane.  Every lane ends with the same bits (each step is one commutative operation on the same two values).  All 64
// lanes must be active.
template <int CTRL>
__device__ __forceinline__ float wave_dpp(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <bool MIN>
__device__ __forceinline__ float wave_reduce(float v)
{
#define MPPI_RED(A, B) (MIN ? fminf((A), (B)) : (A) + (B))
  v = MPPI_RED(v, wave_dpp<0xB1>(v));   // quad_perm [1,0,3,2]
  v = MPPI_RED(v, wave_dpp<0x4E>(v));   // quad_perm [2,3,0,1]
  v = MPPI_RED(v, wave_dpp<0x141>(v));  // row_half_mirror
  v = MPPI_RED(v, wave_dpp<0x140>(v));  // row_mirror
  const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = MPPI_RED(__uint_as_float(x[0]), __uint_as_float(x[1]));
  const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = MPPI_RED(__uint_as_float(y[0]), __uint_as_float(y[1]));
#undef MPPI_RED
  return v;
}
// Called by a whole wave (all 64 lanes active): `mine` = the cost the lane stored, +inf in a lane that stored none.
__device__ __forceinline__ void publish_min_cost(const RolloutArgs &a, const float mine)
{
  if (a.min_cost == nullptr) return;
  const float m = wave_reduce<true>((mine == mine) ? mine : INFINITY);
  if ((threadIdx.x & 63) == 0 && m < INFINITY) {
    const unsigned long long key = ((unsigned long long)(~a.min_cost_tag) << 32) | cost_order_bits(m);
    (void)__hip_atomic_fetch_min(a.min_cost + (blockIdx.x % kMinCostLines) * kMinCostStride, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// The tail's side: true and beta when the keys hold this launch's minimum.
__device__ __forceinline__ bool load_min_cost(const unsigned long long *min_cost, const unsigned tag, float &beta)
{
  if (min_cost == nullptr) return false;
  unsigned long long k[kMinCostLines];
#pragma unroll
  for (int i = 0; i < kMinCostLines; i++) k[i] = __hip_atomic_load(min_cost + i * kMinCostStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long best = k[0];
#pragma unroll
  for (int i = 1; i < kMinCostLines; i++) best = (k[i] < best) ? k[i] : best;
  beta = cost_from_order_bits((unsigned)best);
  return (unsigned)(best >> 32) == ~tag;
}

// Argument block of the batched rollout kernels: grid (groups of the largest instance, instances), workgroup (x, y) runs
// group x of inst[y]: every kernel indexes an instance's rollouts by blockIdx.x alone.
constexpr int kMaxBatch = 4;
template <int NB>
struct QuadBatchArgsT {
  int n;
  RolloutArgs inst[NB];
};
using QuadBatchArgs = QuadBatchArgsT<kMaxBatch>;  // what the ABI layer fills; the launchers pass the two-instance form when n <= 2
// (the two controllers of a tick do not carry two unused blocks through the kernel-argument segment, nor two unused branches
// through the instruction cache: launch call 4.1 -> 3.9 us for the rollout, 3.3 -> 3.0 us for the tail, host profile of
// profiles/r04_t_hostprof_batch.txt)
template <int NB>
inline QuadBatchArgsT<NB> batch_args_prefix(const QuadBatchArgs &b)
{
  QuadBatchArgsT<NB> s;
  s.n = b.n;
  for (int i = 0; i < NB; i++) s.inst[i] = b.inst[i];
  return s;
}
// The argument block of instance blockIdx.y of a batched launch, handed to BODY at a COMPILE-TIME position of the
// kernel-argument segment: the body then reads its parameters exactly as the single-instance kernel does (scalar loads with
// immediate offsets into scalar registers).  With a run-time index the compiler either re-reads the segment inside the
// waves' loops (a reference) or parks a private copy of the block in SCRATCH and turns every pointer loaded from it into a
// flat access (a copy): round 4 found the batched row kernel at 44.8 us beside 33.8 us for the same work launched alone.
static_assert(kMaxBatch == 4, "MPPI_BATCH_DISPATCH");
#define MPPI_BATCH_DISPATCH(NB, B, BODY)     \
  do {                                      \
    if constexpr ((NB) == 2) {              \
      if (blockIdx.y == 0) BODY((B).inst[0]); \
      else BODY((B).inst[1]);               \
    } else {                                \
      switch ((int)blockIdx.y) {            \
        case 0: BODY((B).inst[0]); break;   \
        case 1: BODY((B).inst[1]); break;   \
        case 2: BODY((B).inst[2]); break;   \
        default: BODY((B).inst[3]); break;  \
      }                                     \
    }                                       \
  } while (0)

// Thresholds for the reference's float-vs-double-literal comparisons, as floats:
//   (double)x > 1.57   <=>  x >= kRollCrash   (costs.cu:302)
//   (double)x > 0.001  <=>  x >= kMinSpeed    (costs.cu:340)
//   (double)x > 1e12   <=>  x >= kCostCapGt   (costs.cu:405); replacement value (float)1e12
__device__ constexpr float kRollCrash = 1.57000005245208740234375f;
__device__ constexpr float kMinSpeed = 0.001000000047497451305389404296875f;
__device__ constexpr float kCostCapGt = 1000000061440.0f;
__device__ constexpr float kCostCap = 999999995904.0f;

__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
  // enforceConstraints, neural_net_model.cu:311-323: if (v < lo) v = lo; else if (v > hi) v = hi;
  // as two selects (lo wins when both hold; NaN passes through unchanged)
  float r = (v > hi) ? hi : v;
  r = (v < lo) ? lo : r;
  return r;
}

// tanh for the hidden layers (MPPI_NNET_NONLINEARITY, neural_net_model.cu:35).
// 1 - 2/(exp(2x)+1) on the transcendental unit: |err| <= ~2e-7 absolute over the real line,
// saturates correctly (+-1) for large |x|, keeps NaN.
__device__ __forceinline__ float tanh_fast(float x)
{
  const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);  // exp(2x)
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return fmaf(-2.0f, r, 1.0f);
}

// tanh(z + b) with the bias pre-scaled on the host side of the loop: bs = b * 2*log2(e), so that the
// exp2 argument is ONE fma of the layer's dot product (z + b is never rounded separately; the
// difference to tanh_fast(z + b) is one rounding of the exp2 argument, below its own error).
constexpr float kTanhScale = 2.88539008177792681472f;
__device__ __forceinline__ float tanh_bias(float z, float bs)
{
  const float e = __builtin_amdgcn_exp2f(fmaf(z, kTanhScale, bs));
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return fmaf(-2.0f, r, 1.0f);
}

// Two at a time: the three multiply-adds become packed instructions (v_pk_fma_f32 / v_pk_add_f32,
// one issue slot for two values); exp2 and rcp stay scalar (quarter-rate unit).  Same arithmetic per
// element as tanh_bias.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2 tanh_bias2(f32x2 z, f32x2 bs)
{
  const f32x2 y = __builtin_elementwise_fma(z, f32x2{kTanhScale, kTanhScale}, bs);
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(y.x);
  e.y = __builtin_amdgcn_exp2f(y.y);
  const f32x2 d = e + f32x2{1.0f, 1.0f};
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r, f32x2{1.0f, 1.0f});
}

// sin/cos for the kinematics and the track-cost look-ahead points (neural_net_model.cu:348-349,
// costs.cu:364-367; the reference uses cosf/sinf for the former and __cosf/__sinf for the latter,
// Q7 -- one precise pair serves both here).  Reduction by pi/2 in double (two-constant, exact for
// |x| < 2^30) + degree-7/6 float polynomials: <= 1.3e-7 absolute error, branch-free.
__device__ __forceinline__ void sincos_fast(float x, float &sn, float &cs)
{
  const double xd = (double)x;
  const double q = rint(xd * 0.63661977236758134308);
  double rd = fma(-q, 1.5707963267948966, xd);
  rd = fma(-q, 6.123233995736766e-17, rd);
  const float r = (float)rd;
  const float r2 = r * r;
  const float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
  const float s0 = fmaf(r * r2, ps, r);
  const float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
  const float c0 = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const int qi = (int)q;
  const float a = (qi & 1) ? c0 : s0;
  const float b = (qi & 1) ? s0 : c0;
  sn = (qi & 2) ? -a : a;
  cs = ((qi + 1) & 2) ? -b : b;
}

// coorTransform (costs.cu:351-357) + point/clamp/normalised tex2D addressing (costs.cu:128-154):
// index of the texel that tex2D<float4>(tex, u/w, v/w) returns.
template <bool AFFINE>
__device__ __forceinline__ unsigned texel_index(const CostArgs &c, float x, float y)
{
  float u = fmaf(c.r_c1[0], x, c.r_c2[0] * y) + c.trs[0];
  float v = fmaf(c.r_c1[1], x, c.r_c2[1] * y) + c.trs[1];
  if (!AFFINE) {  // AFFINE: w == 1 exactly, u/w == u
    const float w = fmaf(c.r_c1[2], x, c.r_c2[2] * y) + c.trs[2];
    u = u / w;
    v = v / w;
  }
  float fi = floorf(u * (float)c.map_w);
  float fj = floorf(v * (float)c.map_h);
  fi = fminf(fmaxf(fi, 0.0f), (float)(c.map_w - 1));  // fmaxf(NaN, 0) = 0
  fj = fminf(fmaxf(fj, 0.0f), (float)(c.map_h - 1));
  return (unsigned)((int)fj * c.map_w + (int)fi);
}

// First half of getTrackCost (costs.cu:359-377): the two costmap fetches (front and back of the
// car).  Issued early in the step so that their latency hides under the NN layers.
template <bool AFFINE>
__device__ __forceinline__ void track_fetch(const CostArgs &c, const float *s, float cpsi, float spsi,
                                            float &tf, float &tb)
{
  const float xf = fmaf(0.5f, cpsi, s[0]), yf = fmaf(0.5f, spsi, s[1]);
  const float xb = fmaf(-0.5f, cpsi, s[0]), yb = fmaf(-0.5f, spsi, s[1]);
  tf = c.map[texel_index<AFFINE>(c, xf, yf)];
  tb = c.map[texel_index<AFFINE>(c, xb, yb)];
}

// MPPICosts::computeCost (costs.cu:396-409) in two halves, so that the half that needs no
// costmap texel can run a network layer earlier than the half that does.
struct CostTerms {
  float control_cost, speed_cost, stabilizing_cost;
};

// control (:307-313), speed (:315-326) and stabilizing (:337-349) terms.  s4/s5 = u_x, u_y of
// the state being costed.
template <bool CTRL_COST>
__device__ __forceinline__ void cost_terms_a(const CostArgs &c, const float nu[2], float s4, float s5,
                                             float u0, float u1, float du0, float du1, CostTerms &o)
{
  float control_cost = 0.0f;
  if (CTRL_COST) {  // exactly +0 when both coefficients are 0 and nu is finite and non-zero
    control_cost += c.steering_coeff * du0 * (u0 - du0) / (nu[0] * nu[0]);
    control_cost += c.throttle_coeff * du1 * (u1 - du1) / (nu[1] * nu[1]);
  }
  o.control_cost = control_cost;
  const float err = s4 - c.desired_speed;
  o.speed_cost = c.speed_coeff * (c.l1_cost ? fabsf(err) : err * err);
  // computed unconditionally, selected afterwards (the reference guards with |s4| > 0.001)
  const float slip = -atanf(s5 / fabsf(s4));
  float stab = c.slip_penalty * (slip * slip);
  stab = (fabsf(slip) > c.max_slip_ang) ? stab + c.crash_coeff : stab;
  o.stabilizing_cost = (fabsf(s4) >= kMinSpeed) ? stab : 0.0f;
}

// track term (:379-393), crash flag and crash term (:402, :328-335), the sum in the reference's
// order and the 1e12 / NaN cap (:404-407).  crash is the sticky flag (0/1), updated in place.
__device__ __forceinline__ float cost_terms_b(const CostArgs &c, const CostTerms &a, float tf, float tb,
                                              int &crash)
{
  float track_cost = (fabsf(tf) + fabsf(tb)) * 0.5f;  // == (float)((double)(..)/2.0)
  track_cost = (fabsf(track_cost) < c.track_slop) ? 0.0f : c.track_coeff * track_cost;
  crash |= (int)(tf >= c.boundary_threshold) | (int)(tb >= c.boundary_threshold);
  const float crash_cost = (crash > 0) ? c.crash_cost_discounted : 0.0f;
  float cost = a.control_cost + a.speed_cost + crash_cost + track_cost + a.stabilizing_cost;
  cost = (cost >= kCostCapGt || cost != cost) ? kCostCap : cost;
  return cost;
}

template <bool CTRL_COST>
__device__ __forceinline__ float cost_finish(const CostArgs &c, const float nu[2], float s4, float s5,
                                             float tf, float tb, float u0, float u1, float du0,
                                             float du1, int &crash)
{
  CostTerms t;
  cost_terms_a<CTRL_COST>(c, nu, s4, s5, u0, u1, du0, du1, t);
  return cost_terms_b(c, t, tf, tb, crash);
}

// running_cost += (cost - running_cost)/(1.0*i), in double (mppi_controller.cu:165, Q5).
// The IEEE double division by the integer i is done as Markstein's refinement with the
// correctly rounded reciprocal rt = RN(1/i) from a host table: q0 = d*rt, e = fma(-q0, i, d),
// q = fma(e, rt, q0) is the correctly rounded d/i (3 f64 ops instead of a ~16-op divide).
__device__ __forceinline__ float running_mean(float J, float c, int t, double rt)
{
  const double d = (double)(c - J);
  const double td = (double)t;
  const double q0 = d * rt;
  const double e = fma(-q0, td, d);
  const double q = fma(e, rt, q0);
  return (float)((double)J + q);
}

// ---- hand-over between the wavefronts of a workgroup through LDS sequence words ----
// Every wait draws on the wave's budget of polls (RolloutArgs::spin_budget, kSpinBudget + 64 T by default).
// A wave whose budget runs out stops waiting for good -- a loud failure instead of a hung GPU: it carries
// on with whatever the LDS holds (so the other waves never starve because of it) and, when it is through
// its T steps, raises the WORKGROUP'S FAIL WORD and then its own "finished" word.  The cost wave, the one
// that writes the results, waits for the finished words of all other waves of the group (that costs it
// nothing: the kernel cannot end before they do) and poisons the costs of the whole group with NaN if the
// fail word is up or its own budget ran out -- whichever wave starved.  The host turns the NaN into
// MPPI_ERR_HIP (eta is not >= 1).
constexpr int kSpinBudget = 1 << 22;

// The hand-over instructions are written as ds_* assembly: they must reach the LDS in exactly this
// order (data, then sequence word / sequence word, then data) and must not carry the waits the
// compiler attaches to volatile accesses.  "memory" clobbers keep the ordinary LDS accesses
// (records) on their side of a hand-over.
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ void lds_publish(uint32_t addr, int v)
{
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_put4(uint32_t addr, f32x4 v)
{
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// one word, wave-uniform
__device__ __forceinline__ int lds_peek(uint32_t addr)
{
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
// The layer-0 B operand of k-step 0, [s3, s4, s5, s6][g] (g = lane >> 4), as a select tree of depth 2 (three
// v_cndmask on lane-constant conditions; the ternary CHAIN on g is one level deeper on the T-step chain).
__device__ __forceinline__ float row_sel(int g, float s3, float s4, float s5, float s6)
{
  const float lo = (g == 0) ? s3 : s4;
  const float hi = (g == 2) ? s5 : s6;
  return (g < 2) ? lo : hi;
}

// one word per lane, not made uniform
__device__ __forceinline__ int lds_peek_lanes(uint32_t addr)
{
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}

// Budget of one wavefront's waits (all of them together): `while (<not yet there> && --budget > 0) poll;`.
// fault_wave (tests) starts one role with an exhausted budget: it then never waits for anybody, which is
// exactly what a wave that gave up does, and is reported like one.
__device__ __forceinline__ int spin_budget_init(int spin_budget, int T, bool inject_fault)
{
  if (inject_fault) return 0;
  return (spin_budget > 0 ? spin_budget : kSpinBudget) + 64 * T;
}
// Last thing a wave does: raise the fail word if any of its waits gave up, then its "finished" word.
__device__ __forceinline__ void spin_finish(int budget, uint32_t a_fail, uint32_t a_fin)
{
  if (budget <= 0) lds_publish(a_fail, 1);
  lds_publish(a_fin, 1);
}

// Gated launch (the chained control ticks of abi_solve.hip; SH with gstate[8] and gate_open[8]: the row, m44 and multi4-tree forms): ONE wave of the group -- the pose wave; the multi form: the control wave -- waits for the host to open the gate -- word 7 of this workgroup's copy of the gate block equal to
// a.gate_seq -- and hands the block's vehicle state to the dynamics waves through LDS.  The gate word is host-written memory
// (device memory the host stores into through the PCIe BAR, or host-mapped memory): system-scope loads.  The wait is bounded by
// the 100 MHz real-time counter (100 ms); a gate that stays shut, or is opened with the cancel bit, leaves the pose wave with an
// exhausted poll budget: the group's costs are poisoned (NaN) as after any other failed hand-over, the kernel ends.
template <class SH>
__device__ __forceinline__ int group_gate_wait(const RolloutArgs &a, SH &sh)
{
  const int lane = threadIdx.x & 63;
  const unsigned *blk = a.gate + (size_t)((int)blockIdx.x % kGateReplicas) * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned v = 0;
  for (;;) {
    v = __hip_atomic_load(blk + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((v & ~kGateCancel) == a.gate_seq) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > 10000000ull) { v = kGateCancel; break; }
    __builtin_amdgcn_s_sleep(2);
  }
  // the state words were stored before the gate word (the host fences between them): loaded only now
  asm volatile("" ::: "memory");
  const float sv = __uint_as_float(__hip_atomic_load(blk + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  if (lane < 7) sh.gstate[lane] = sv;
  lds_publish(lds_addr(&sh.gate_open[0]), 1);
  return (v & kGateCancel) ? 1 : 0;
}


}  // namespace mppi
