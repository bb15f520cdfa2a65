// row_bcast_ub.hip -- the 6-32-32-4 recurrence of the row form (rollout_row.hip) with the activations of a layer handed
// round INSIDE THE REGISTERS: a rollout is one 16-lane DPP row, and `row_newbcast:q` gives every lane of a row the value of
// lane q -- one v_mov_b32_dpp per k (off the dependent chain) instead of an LDS round trip per layer.  Measured alone on a
// SIMD, against the LDS form of tools/ub/row_lds_ub.hip, bit for bit the same recurrence:
//   form A: hidden layers and the output layer as v_mov_b32_dpp + v_pk_fma_f32 (a lane owns two neurons / two outputs);
//   form B: the output layer as ONE chain per lane (lane p -> output p & 3) of v_fmac_f32_dpp, the broadcast inside the
//           multiply-add (inline assembly: the compiler does not fold the move into the VOP2 form by itself).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ub/row_bcast_ub.hip -o row_bcast_ub
#include "../../autorally_amd/csrc/mppi_device.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace mppi;

constexpr int H = 32;

struct RowWeights {
  f32x2 w1[6], w2[H], w3[H];
  f32x2 b1s, b2s, b3;
};

__device__ __forceinline__ void row_load(const float *theta, int p, RowWeights &W)
{
  const float *W1 = theta, *B1 = W1 + H * 6, *W2 = B1 + H, *B2 = W2 + H * H, *W3 = B2 + H, *B3 = W3 + 4 * H;
  const int j0 = 2 * p, j1 = 2 * p + 1, o0 = 2 * (p & 1), o1 = o0 + 1;
#pragma unroll
  for (int k = 0; k < 6; k++) W.w1[k] = f32x2{W1[j0 * 6 + k], W1[j1 * 6 + k]};
#pragma unroll
  for (int k = 0; k < H; k++) W.w2[k] = f32x2{W2[j0 * H + k], W2[j1 * H + k]};
#pragma unroll
  for (int k = 0; k < H; k++) W.w3[k] = f32x2{W3[o0 * H + k], W3[o1 * H + k]};
  W.b1s = f32x2{B1[j0] * kTanhScale, B1[j1] * kTanhScale};
  W.b2s = f32x2{B2[j0] * kTanhScale, B2[j1] * kTanhScale};
  W.b3 = f32x2{B3[o0], B3[o1]};
}

struct RowLds {
  float act[2][4][H];
};

template <int Q>
__device__ __forceinline__ float bc(float a)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x150 + Q, 0xF, 0xF, false));  // row_newbcast:Q
}

// z = sum_k w[k] * a[k], lane q of the row holds a[2q], a[2q+1]; k ascending.  The schedule of the product
// (rollout_row.hip: row_dot_bc): the move for k+1, then the multiply-add of k, held there by scheduling barriers.
template <int K>
__device__ __forceinline__ float bc_k(f32x2 a)
{
  return bc<(K >> 1)>((K & 1) ? a.y : a.x);
}
template <int K>
__device__ __forceinline__ void dot_step(f32x2 &z, float &v, const f32x2 *w, f32x2 a)
{
  const float vn = bc_k<(K + 1 < 32 ? K + 1 : 31)>(a);
  z = __builtin_elementwise_fma(w[K], f32x2{v, v}, z);
  __builtin_amdgcn_sched_barrier(0);
  v = vn;
}
__device__ __forceinline__ f32x2 row_dot_bc(const f32x2 *w, f32x2 a)
{
  f32x2 z = {0.0f, 0.0f};
  float v = bc_k<0>(a);
  __builtin_amdgcn_sched_barrier(0);
#define RD4(K) dot_step<K>(z, v, w, a); dot_step<K + 1>(z, v, w, a); dot_step<K + 2>(z, v, w, a); dot_step<K + 3>(z, v, w, a);
  RD4(0) RD4(4) RD4(8) RD4(12) RD4(16) RD4(20) RD4(24) RD4(28)
#undef RD4
  return z;
}

// ---- LDS form (reference of this file: the product's step, tools/ub/row_lds_ub.hip k_row_dpp) ----
__global__ __launch_bounds__(256) void k_lds(const float *theta, float *out, unsigned long long *cyc, int iters, float dt)
{
  __shared__ __attribute__((aligned(16))) RowLds lds[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane >> 4, p = lane & 15;
  const bool odd = (p & 1) != 0;
  RowWeights W;
  row_load(theta, p, W);
  RowLds &L = lds[w];
  const int gk = (blockIdx.x * 4 + w) * 4 + r;
  f32x2 sp = odd ? f32x2{0.1f, 0.0f} : f32x2{0.01f * (float)(gk % 64), 5.0f};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    f32x2 so;
    so.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sp.x), 0xB1, 0xF, 0xF, false));
    so.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sp.y), 0xB1, 0xF, 0xF, false));
    const f32x2 slo = odd ? so : sp, shi = odd ? sp : so;
    f32x2 z = {0.0f, 0.0f};
    z = __builtin_elementwise_fma(W.w1[0], f32x2{slo.x, slo.x}, z);
    z = __builtin_elementwise_fma(W.w1[1], f32x2{slo.y, slo.y}, z);
    z = __builtin_elementwise_fma(W.w1[2], f32x2{shi.x, shi.x}, z);
    z = __builtin_elementwise_fma(W.w1[3], f32x2{shi.y, shi.y}, z);
    z = __builtin_elementwise_fma(W.w1[4], f32x2{0.1f, 0.1f}, z);
    z = __builtin_elementwise_fma(W.w1[5], f32x2{0.1f, 0.1f}, z);
    *reinterpret_cast<f32x2 *>(&L.act[0][r][2 * p]) = tanh_bias2(z, W.b1s);
    __builtin_amdgcn_wave_barrier();
    {
      float4 v[H / 4];
#pragma unroll
      for (int q = 0; q < H / 4; q++) v[q] = *reinterpret_cast<const float4 *>(&L.act[0][r][4 * q]);
      z = f32x2{0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < H / 4; q++) {
        z = __builtin_elementwise_fma(W.w2[4 * q + 0], f32x2{v[q].x, v[q].x}, z);
        z = __builtin_elementwise_fma(W.w2[4 * q + 1], f32x2{v[q].y, v[q].y}, z);
        z = __builtin_elementwise_fma(W.w2[4 * q + 2], f32x2{v[q].z, v[q].z}, z);
        z = __builtin_elementwise_fma(W.w2[4 * q + 3], f32x2{v[q].w, v[q].w}, z);
      }
      *reinterpret_cast<f32x2 *>(&L.act[1][r][2 * p]) = tanh_bias2(z, W.b2s);
    }
    __builtin_amdgcn_wave_barrier();
    {
      float4 v[H / 4];
#pragma unroll
      for (int q = 0; q < H / 4; q++) v[q] = *reinterpret_cast<const float4 *>(&L.act[1][r][4 * q]);
      z = f32x2{0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < H / 4; q++) {
        z = __builtin_elementwise_fma(W.w3[4 * q + 0], f32x2{v[q].x, v[q].x}, z);
        z = __builtin_elementwise_fma(W.w3[4 * q + 1], f32x2{v[q].y, v[q].y}, z);
        z = __builtin_elementwise_fma(W.w3[4 * q + 2], f32x2{v[q].z, v[q].z}, z);
        z = __builtin_elementwise_fma(W.w3[4 * q + 3], f32x2{v[q].w, v[q].w}, z);
      }
      sp = __builtin_elementwise_fma(z + W.b3, f32x2{dt, dt}, sp);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + w] = c1 - c0;
  if (p < 2) { out[gk * 4 + 2 * p] = sp.x; out[gk * 4 + 2 * p + 1] = sp.y; }
}

// ---- form A: every layer as mov_dpp + pk_fma; state pair as in the product (lane pair (p, p^1) holds the whole state) ----
// SC: the two tanh of a lane's pair as two scalar chains (fma, exp, add, rcp, fma each) instead of the packed form: the first
// activation of the next layer's chain is ready one transcendental issue earlier
template <bool SC>
__device__ __forceinline__ f32x2 tanh2(f32x2 z, f32x2 b)
{
  if (SC) return f32x2{tanh_bias(z.x, b.x), tanh_bias(z.y, b.y)};
  return tanh_bias2(z, b);
}
template <bool SC>
__global__ __launch_bounds__(256) void k_bc_a(const float *theta, float *out, unsigned long long *cyc, int iters, float dt)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane >> 4, p = lane & 15;
  const bool odd = (p & 1) != 0;
  RowWeights W;
  row_load(theta, p, W);
  const int gk = (blockIdx.x * 4 + w) * 4 + r;
  f32x2 sp = odd ? f32x2{0.1f, 0.0f} : f32x2{0.01f * (float)(gk % 64), 5.0f};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    // lane 0 of the row holds (s3, s4), lane 1 (s5, s6)
    const float s3 = bc<0>(sp.x), s4 = bc<0>(sp.y), s5 = bc<1>(sp.x), s6 = bc<1>(sp.y);
    f32x2 z = {0.0f, 0.0f};
    z = __builtin_elementwise_fma(W.w1[0], f32x2{s3, s3}, z);
    z = __builtin_elementwise_fma(W.w1[1], f32x2{s4, s4}, z);
    z = __builtin_elementwise_fma(W.w1[2], f32x2{s5, s5}, z);
    z = __builtin_elementwise_fma(W.w1[3], f32x2{s6, s6}, z);
    z = __builtin_elementwise_fma(W.w1[4], f32x2{0.1f, 0.1f}, z);
    z = __builtin_elementwise_fma(W.w1[5], f32x2{0.1f, 0.1f}, z);
    const f32x2 a0 = tanh2<SC>(z, W.b1s);
    const f32x2 a1 = tanh2<SC>(row_dot_bc(W.w2, a0), W.b2s);
    z = row_dot_bc(W.w3, a1);
    sp = __builtin_elementwise_fma(z + W.b3, f32x2{dt, dt}, sp);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + w] = c1 - c0;
  if (p < 2) { out[gk * 4 + 2 * p] = sp.x; out[gk * 4 + 2 * p + 1] = sp.y; }
}

// ---- form P: form A with the per-step bookkeeping of the product's dynamics wave (rollout_row.hip): the state record and the
// sequence word to LDS, the next controls and the two progress words from LDS, the scalar end-of-step test ----
struct BookLds {
  float rec[16][16][4];
  float ctl[16][16][4];
  int seq[64], pub[64], done[64];
};
template <int EXTRAS>
__global__ __launch_bounds__(256) void k_bc_p(const float *theta, float *out, unsigned long long *cyc, int iters, float dt)
{
  __shared__ __attribute__((aligned(16))) BookLds L;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane >> 4, p = lane & 15;
  const int jr = 4 * w + r;
  const bool odd = (p & 1) != 0;
  RowWeights W;
  row_load(theta, p, W);
  const int gk = (blockIdx.x * 4 + w) * 4 + r;
  if (w == 0) { L.pub[lane] = 1 << 30; L.done[lane] = 1 << 30; }
  for (int i = threadIdx.x; i < 16 * 16 * 4; i += 256) (&L.ctl[0][0][0])[i] = 0.1f;
  __syncthreads();
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile float __attribute__((address_space(3))) *lds_float_p;
  const lds_int_p p_pub = (lds_int_p)&L.pub[0];
  const lds_int_p p_cd = (lds_int_p)&L.done[0];
  const lds_float_p p_u = (lds_float_p)&L.ctl[0][jr][0];
  const uint32_t a_myseq = lds_addr(&L.seq[lane]);
  f32x2 sp = odd ? f32x2{0.1f, 0.0f} : f32x2{0.01f * (float)(gk % 64), 5.0f};
  float u0n = 0.1f, u1n = 0.1f;
  int budget = 1 << 20;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < iters; t++) {
    const int slot = t & 15;
    const float u0 = u0n, u1 = u1n;
    f32x2 so;
    so.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sp.x), 0xB1, 0xF, 0xF, false));
    so.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sp.y), 0xB1, 0xF, 0xF, false));
    const f32x2 slo = odd ? so : sp, shi = odd ? sp : so;
    if (EXTRAS >= 1) {
      if (p < 2) *reinterpret_cast<f32x2 *>(&L.rec[slot][jr][2 * p]) = sp;
      lds_publish(a_myseq, t + 1);
    }
    f32x2 z = {0.0f, 0.0f};
    z = __builtin_elementwise_fma(W.w1[0], f32x2{slo.x, slo.x}, z);
    z = __builtin_elementwise_fma(W.w1[1], f32x2{slo.y, slo.y}, z);
    z = __builtin_elementwise_fma(W.w1[2], f32x2{shi.x, shi.x}, z);
    z = __builtin_elementwise_fma(W.w1[3], f32x2{shi.y, shi.y}, z);
    z = __builtin_elementwise_fma(W.w1[4], f32x2{u0, u0}, z);
    z = __builtin_elementwise_fma(W.w1[5], f32x2{u1, u1}, z);
    const int sn = ((t + 1) & 15) * 64;
    int cp_v = 1 << 30, cd_v = 1 << 30;
    float un0_v = 0.1f, un1_v = 0.1f;
    if (EXTRAS >= 2) {
      cp_v = *p_pub;
      un0_v = p_u[sn]; un1_v = p_u[sn + 1];
      cd_v = *p_cd;
    }
    const f32x2 a0 = tanh_bias2(z, W.b1s);
    const f32x2 a1 = tanh_bias2(row_dot_bc(W.w2, a0), W.b2s);
    z = row_dot_bc(W.w3, a1);
    sp = __builtin_elementwise_fma(z + W.b3, f32x2{dt, dt}, sp);
    asm volatile("" : "+v"(sp));
    if (EXTRAS >= 3) {
      int cp = __builtin_amdgcn_readfirstlane(cp_v), cd = __builtin_amdgcn_readfirstlane(cd_v);
      while (((cp < t + 2) | (cd < t - 14)) && --budget > 0) {
        cp = __builtin_amdgcn_readfirstlane(*p_pub);
        un0_v = p_u[sn]; un1_v = p_u[sn + 1];
        cd = __builtin_amdgcn_readfirstlane(*p_cd);
      }
    }
    u0n = un0_v; u1n = un1_v;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + w] = c1 - c0;
  if (p < 2) { out[gk * 4 + 2 * p] = sp.x + (float)(budget & 0); out[gk * 4 + 2 * p + 1] = sp.y; }
}

// ---- form Q: the bookkeeping rearranged around a chain without LDS waits: the record of every lane (lanes p >= 2 into a
// dump row: no exec masking), the next controls and the progress words requested at the TOP of the step (complete before
// layer 1, whose packed multiply-adds formally read the odd halves of the move registers -- where the register allocator
// likes to put pending LDS results), the scalar test in front of the output layer; SCHED: 0 = scheduling barriers per k,
// 1 = value pins per k (other instructions may move), 2 = the compiler's own schedule ----
template <int K, int SCHED>
__device__ __forceinline__ void dot_step_q(f32x2 &z, float &v, const f32x2 *w, f32x2 &a)
{
  const float vn = bc_k<(K + 1 < 32 ? K + 1 : 31)>(a);
  z = __builtin_elementwise_fma(w[K], f32x2{v, v}, z);
  if (SCHED == 0) __builtin_amdgcn_sched_barrier(0);
  if (SCHED == 1) asm volatile("" : "+v"(z), "+v"(a));
  v = vn;
}
template <int SCHED>
__device__ __forceinline__ f32x2 row_dot_q(const f32x2 *w, f32x2 a)
{
  f32x2 z = {0.0f, 0.0f};
  float v = bc_k<0>(a);
  if (SCHED == 0) __builtin_amdgcn_sched_barrier(0);
#define RD4(K) dot_step_q<K, SCHED>(z, v, w, a); dot_step_q<K + 1, SCHED>(z, v, w, a); dot_step_q<K + 2, SCHED>(z, v, w, a); dot_step_q<K + 3, SCHED>(z, v, w, a);
  RD4(0) RD4(4) RD4(8) RD4(12) RD4(16) RD4(20) RD4(24) RD4(28)
#undef RD4
  return z;
}
struct BookLdsQ {
  float rec[16][16][4];
  float ctl[16][16][4];
  int seq[64], pub[64], done[64];
  float dump[4][64][2];
};
template <int SCHED>
__global__ __launch_bounds__(256) void k_bc_q(const float *theta, float *out, unsigned long long *cyc, int iters, float dt)
{
  __shared__ __attribute__((aligned(16))) BookLdsQ L;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane >> 4, p = lane & 15;
  const int jr = 4 * w + r;
  const bool odd = (p & 1) != 0;
  RowWeights W;
  row_load(theta, p, W);
  const int gk = (blockIdx.x * 4 + w) * 4 + r;
  if (w == 0) { L.pub[lane] = 1 << 30; L.done[lane] = 1 << 30; }
  for (int i = threadIdx.x; i < 16 * 16 * 4; i += 256) (&L.ctl[0][0][0])[i] = 0.1f;
  __syncthreads();
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile float __attribute__((address_space(3))) *lds_float_p;
  const lds_int_p p_pub = (lds_int_p)&L.pub[0];
  const lds_int_p p_cd = (lds_int_p)&L.done[0];
  const lds_float_p p_u = (lds_float_p)&L.ctl[0][jr][0];
  const uint32_t a_myseq = lds_addr(&L.seq[lane]);
  // record address of this lane in ring slot 0: the real record for p < 2, a dump row (never read) for the others
  const uint32_t a_rec0 = (p < 2) ? lds_addr(&L.rec[0][jr][2 * p]) : lds_addr(&L.dump[w][lane][0]);
  const uint32_t rec_stride = (p < 2) ? 256u : 0u;
  f32x2 sp = odd ? f32x2{0.1f, 0.0f} : f32x2{0.01f * (float)(gk % 64), 5.0f};
  float u0n = 0.1f, u1n = 0.1f;
  int budget = 1 << 20;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < iters; t++) {
    const int slot = t & 15;
    const float u0 = u0n, u1 = u1n;
    asm volatile("ds_write_b64 %0, %1" ::"v"(a_rec0 + (uint32_t)slot * rec_stride), "v"(sp) : "memory");
    lds_publish(a_myseq, t + 1);
    const int sn = ((t + 1) & 15) * 64;
    // requested here, by hand (the scheduler would move tracked loads down to their use and wait there)
    int cp_v, cd_v;
    float un0_v, un1_v;
    asm volatile("ds_read_b32 %0, %5\n\tds_read_b32 %1, %6\n\tds_read_b32 %2, %6 offset:4\n\tds_read_b32 %3, %7"
                 : "=&v"(cp_v), "=&v"(un0_v), "=&v"(un1_v), "=&v"(cd_v), "+v"(sp)  // sp: layer 0 stays behind the requests
                 : "v"(lds_addr((const void *)p_pub)), "v"(lds_addr((const void *)(p_u + sn))), "v"(lds_addr((const void *)p_cd))
                 : "memory");
    // lane 0 of the row holds (s3, s4), lane 1 (s5, s6)
    const float s3 = bc<0>(sp.x), s4 = bc<0>(sp.y), s5 = bc<1>(sp.x), s6 = bc<1>(sp.y);
    f32x2 z = {0.0f, 0.0f};
    z = __builtin_elementwise_fma(W.w1[0], f32x2{s3, s3}, z);
    z = __builtin_elementwise_fma(W.w1[1], f32x2{s4, s4}, z);
    z = __builtin_elementwise_fma(W.w1[2], f32x2{s5, s5}, z);
    z = __builtin_elementwise_fma(W.w1[3], f32x2{s6, s6}, z);
    z = __builtin_elementwise_fma(W.w1[4], f32x2{u0, u0}, z);
    z = __builtin_elementwise_fma(W.w1[5], f32x2{u1, u1}, z);
    f32x2 a0 = tanh_bias2(z, W.b1s);
    // the LDS reads of the top of the step are complete by now: the wait is pinned HERE (behind a0, in front of layer 1)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cp_v), "+v"(un0_v), "+v"(un1_v), "+v"(cd_v), "+v"(a0));
    const int cp_s = __builtin_amdgcn_readfirstlane(cp_v), cd_s = __builtin_amdgcn_readfirstlane(cd_v);
    const bool need_wait = (cp_s < t + 2) | (cd_s < t - 14);
    const f32x2 a1 = tanh_bias2(row_dot_q<SCHED>(W.w2, a0), W.b2s);
    z = row_dot_q<SCHED>(W.w3, a1);
    sp = __builtin_elementwise_fma(z + W.b3, f32x2{dt, dt}, sp);
    asm volatile("" : "+v"(sp));
    if (__builtin_expect(need_wait, 0)) {
      int cp = cp_s, cd = cd_s;
      while (((cp < t + 2) | (cd < t - 14)) && --budget > 0) {
        cp = __builtin_amdgcn_readfirstlane(*p_pub);
        un0_v = p_u[sn]; un1_v = p_u[sn + 1];
        cd = __builtin_amdgcn_readfirstlane(*p_cd);
      }
    }
    u0n = un0_v; u1n = un1_v;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + w] = c1 - c0;
  if (p < 2) { out[gk * 4 + 2 * p] = sp.x + (float)(budget & 0); out[gk * 4 + 2 * p + 1] = sp.y; }
}

// ---- form B: output layer = one v_fmac_f32_dpp chain per lane (lane p -> output p & 3) ----
// z += a(lane Q of the row) * w: the broadcast inside the multiply-add.  NOP: two wait states in front (a VALU write of the
// DPP source needs them; the compiler does not see into the assembly)
template <int Q, bool NOP>
__device__ __forceinline__ void fmac_bc(float &z, float a, float w)
{
  if (NOP)
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(z) : "v"(a), "v"(w), "n"(Q));
  else
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(z) : "v"(a), "v"(w), "n"(Q));
}
template <int Q>
__device__ __forceinline__ void out_step(float &z, const float *w, f32x2 a)
{
  fmac_bc<Q, Q == 0>(z, a.x, w[2 * Q]);
  fmac_bc<Q, false>(z, a.y, w[2 * Q + 1]);
}
template <bool ASM_HIDDEN>
__global__ __launch_bounds__(256) void k_bc_b(const float *theta, float *out, unsigned long long *cyc, int iters, float dt)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane >> 4, p = lane & 15;
  RowWeights W;
  row_load(theta, p, W);
  // output layer: lane p computes output o = p & 3 alone
  const int o = p & 3;
  const float *W3 = theta + H * 6 + H + H * H + H, *B3 = W3 + 4 * H;
  float w3[H];
#pragma unroll
  for (int k = 0; k < H; k++) w3[k] = W3[o * H + k];
  const float b3 = B3[o];
  const int gk = (blockIdx.x * 4 + w) * 4 + r;
  const float s_init[4] = {0.01f * (float)(gk % 64), 5.0f, 0.1f, 0.0f};
  float s = s_init[o];  // state component 3 + o
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    f32x2 z = {0.0f, 0.0f};
    if (ASM_HIDDEN) {
      float zx = 0.0f, zy = 0.0f;
      fmac_bc<0, true>(zx, s, W.w1[0].x); fmac_bc<0, false>(zy, s, W.w1[0].y);
      fmac_bc<1, false>(zx, s, W.w1[1].x); fmac_bc<1, false>(zy, s, W.w1[1].y);
      fmac_bc<2, false>(zx, s, W.w1[2].x); fmac_bc<2, false>(zy, s, W.w1[2].y);
      fmac_bc<3, false>(zx, s, W.w1[3].x); fmac_bc<3, false>(zy, s, W.w1[3].y);
      z = f32x2{zx, zy};
    } else {
      const float s3 = bc<0>(s), s4 = bc<1>(s), s5 = bc<2>(s), s6 = bc<3>(s);
      z = __builtin_elementwise_fma(W.w1[0], f32x2{s3, s3}, z);
      z = __builtin_elementwise_fma(W.w1[1], f32x2{s4, s4}, z);
      z = __builtin_elementwise_fma(W.w1[2], f32x2{s5, s5}, z);
      z = __builtin_elementwise_fma(W.w1[3], f32x2{s6, s6}, z);
    }
    z = __builtin_elementwise_fma(W.w1[4], f32x2{0.1f, 0.1f}, z);
    z = __builtin_elementwise_fma(W.w1[5], f32x2{0.1f, 0.1f}, z);
    const f32x2 a0 = tanh_bias2(z, W.b1s);
    const f32x2 a1 = tanh_bias2(row_dot_bc(W.w2, a0), W.b2s);
    float zo = 0.0f;
    out_step<0>(zo, w3, a1); out_step<1>(zo, w3, a1); out_step<2>(zo, w3, a1); out_step<3>(zo, w3, a1);
    out_step<4>(zo, w3, a1); out_step<5>(zo, w3, a1); out_step<6>(zo, w3, a1); out_step<7>(zo, w3, a1);
    out_step<8>(zo, w3, a1); out_step<9>(zo, w3, a1); out_step<10>(zo, w3, a1); out_step<11>(zo, w3, a1);
    out_step<12>(zo, w3, a1); out_step<13>(zo, w3, a1); out_step<14>(zo, w3, a1); out_step<15>(zo, w3, a1);
    s = fmaf(zo + b3, dt, s);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + w] = c1 - c0;
  if (p < 4) out[gk * 4 + p] = s;
}

// dependent-issue distances of the candidate instructions, one wave per SIMD: cycles per instruction of a chain of N
template <int MODE>
__global__ __launch_bounds__(256) void k_lat(float *out, unsigned long long *cyc, int iters)
{
  float z = threadIdx.x * 1e-3f, a = 0.5f, w = 1.0001f, z2 = 0.25f;
  f32x2 zz = {z, z2};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 32; j++) {
      if (MODE == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(z) : "v"(a), "v"(w));
      if (MODE == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(z) : "v"(a), "v"(w));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(zz) : "v"(f32x2{a, a}), "v"(f32x2{w, w}));
      if (MODE == 3) {  // two interleaved fmac_dpp chains
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(z) : "v"(a), "v"(w));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(z2) : "v"(a), "v"(w));
      }
      if (MODE == 4) {  // mov_dpp beside a dependent pk_fma chain
        float v;
        asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(a));
        zz = __builtin_elementwise_fma(f32x2{w, w}, f32x2{v, v}, zz);
      }
      if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z) : "v"(a), "v"(w));
      if (MODE == 7) {  // mov_dpp beside a dependent plain v_fmac chain
        float v;
        asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(a));
        asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(z) : "v"(v), "v"(w));
      }
      if (MODE == 5) {  // mov_dpp (result dependent on the chain: the DPP read of a fresh VALU result)
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(z));
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
  out[(blockIdx.x * 256 + threadIdx.x) & 16383] = z + z2 + zz.x + zz.y;  // d_o holds 16384 floats
}

int main(int argc, char **argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 100;
  std::vector<float> th(H * 6 + H + H * H + H + 4 * H + 4);
  for (size_t i = 0; i < th.size(); i++) th[i] = 0.25f * (float)((int)((i * 2654435761u) >> 20 & 255) - 128) / 128.0f;
  float *d_t, *d_o; unsigned long long *d_c;
  hipMalloc(&d_t, th.size() * 4); hipMemcpy(d_t, th.data(), th.size() * 4, hipMemcpyHostToDevice);
  const int NB = 256, NR = NB * 16;
  hipMalloc(&d_o, NR * 4 * 4); hipMalloc(&d_c, 4100 * 8);
  const float dt = 0.02f;
  std::vector<float> ref(NR * 4), got(NR * 4);
  auto run = [&](const char *name, auto kern, bool is_ref) {
    hipMemset(d_o, 0, NR * 16);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(kern, dim3(NB), dim3(256), 0, 0, d_t, d_o, d_c, iters, dt);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<unsigned long long> c(NB * 4);
    hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : c) s += (double)v;
    hipMemcpy(is_ref ? ref.data() : got.data(), d_o, NR * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    if (!is_ref) for (int i = 0; i < NR * 4; i++) bad += memcmp(&ref[i], &got[i], 4) != 0;
    printf("%-58s %.0f cycles per step", name, s / c.size() / iters);
    if (!is_ref) printf("   %s (%d of %d words differ from the LDS form)", bad ? "MISMATCH" : "bit-identical", bad, NR * 4);
    printf("\n");
  };
  run("LDS form (product)", k_lds, true);
  run("form A: mov_dpp + pk_fma, all layers", k_bc_a<false>, false);
  run("form A, the pair's two tanh as scalar chains", k_bc_a<true>, false);
  run("form P0: A in the product's loop shape (partner move, u from regs)", k_bc_p<0>, false);
  run("form P1: + state record and sequence word to LDS", k_bc_p<1>, false);
  run("form P2: + next controls and progress words from LDS", k_bc_p<2>, false);
  run("form P3: + scalar end-of-step test (never waits)", k_bc_p<3>, false);
  run("form Q0: bookkeeping rearranged, scheduling barriers per k", k_bc_q<0>, false);
  run("form Q1: bookkeeping rearranged, value pins per k", k_bc_q<1>, false);
  run("form Q2: bookkeeping rearranged, the compiler's schedule", k_bc_q<2>, false);
  run("form B: hidden mov_dpp + pk_fma, output fmac_dpp chain", k_bc_b<false>, false);
  run("form B2: as B, layer 0 by fmac_dpp too", k_bc_b<true>, false);
  const char *names[] = {"v_fmac_f32 dependent", "v_fmac_f32_dpp row_newbcast dependent", "v_pk_fma_f32 dependent",
                         "two interleaved v_fmac_f32_dpp chains (per pair)", "v_mov_b32_dpp + dependent v_pk_fma_f32 (per pair)",
                         "s_nop 1 + v_mov_b32_dpp on its own result", "v_fma_f32 (VOP3) dependent",
                         "v_mov_b32_dpp + dependent v_fmac_f32 (per pair)"};
  for (int m = 0; m < 8; m++) {
    for (int rep = 0; rep < 2; rep++) {
      if (m == 0) hipLaunchKernelGGL(k_lat<0>, dim3(NB), dim3(256), 0, 0, d_o, d_c, 100);
      if (m == 1) hipLaunchKernelGGL(k_lat<1>, dim3(NB), dim3(256), 0, 0, d_o, d_c, 100);
      if (m == 2) hipLaunchKernelGGL(k_lat<2>, dim3(NB), dim3(256), 0, 0, d_o, d_c, 100);
      if (m == 3) hipLaunchKernelGGL(k_lat<3>, dim3(NB), dim3(256), 0, 0, d_o, d_c, 100);
      if (m == 4) hipLaunchKernelGGL(k_lat<4>, dim3(NB), dim3(256), 0, 0, d_o, d_c, 100);
      if (m == 5) hipLaunchKernelGGL(k_lat<5>, dim3(NB), dim3(256), 0, 0, d_o, d_c, 100);
      if (m == 6) hipLaunchKernelGGL(k_lat<6>, dim3(NB), dim3(256), 0, 0, d_o, d_c, 100);
      if (m == 7) hipLaunchKernelGGL(k_lat<7>, dim3(NB), dim3(256), 0, 0, d_o, d_c, 100);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(NB * 4);
    hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : c) s += (double)v;
    printf("%-58s %.2f cycles each\n", names[m], s / c.size() / 100 / 32);
  }
  return 0;
}
