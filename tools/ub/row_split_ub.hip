// row_split_ub.hip -- one 32-input hidden layer of the row form (rollout_row.hip: lane q of a 16-lane DPP row holds the
// activations a[2q], a[2q+1], every lane owns two neurons) as a dependent recurrence (the layer's output is the next
// layer's input), alone on a SIMD.  What does the k-ascending chain cost, and what do two interleaved chains with ONE
// 64-bit broadcast per activation pair cost?
//   form 0: one chain, k ascending, one v_mov_b32_dpp row_newbcast per k (the product's row_dot_bc)
//   form 1: two chains (even k / odd k), still one v_mov_b32_dpp per k
//   form 2: two chains, one v_mov_b64_dpp row_newbcast per PAIR (a[2q], a[2q+1]), the halves picked by op_sel
//   form 3: four chains (k mod 4), one v_mov_b64_dpp per pair
//   form 4: ONE chain, k ascending (the reference's order, bit for bit form 0), one v_mov_b64_dpp per pair
//   form 5: as 4, the move between the two multiply-adds of a pair
//   form 6: form 4 + the layer's tanh (1 - 2 / (exp2(c z + b) + 1): two v_exp_f32, two v_rcp_f32, three packed operations per lane) --
//           the whole hidden layer of the product's dynamics wave: 4 rollouts per wave, one wave per SIMD
//   form 7: VERDICT round 4, item 5 -- TWO waves per SIMD, each 2 rollouts x 32 lanes, lane = neuron pair x k-half: 16 dependent
//           v_pk_fma_f32 + 8 v_mov_b64_dpp per lane, the halves added across the two DPP rows of a rollout (two
//           v_permlane16_swap_b32 + one packed add), the upper row's pairs rotated by 8 lanes (one row-masked v_mov_b64_dpp) so
//           that row_newbcast:i serves both halves, + the tanh (now evaluated by BOTH rows of a rollout); launched with 8 waves
//           per workgroup (two per SIMD): the same 4 rollouts per SIMD and layer as form 6 -- compare cycles per layer
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ub/row_split_ub.hip -o row_split_ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int H = 32;

template <int Q>
__device__ __forceinline__ float bc32(float a)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x150 + Q, 0xF, 0xF, false));
}
template <int Q>
__device__ __forceinline__ f32x2 bc64(f32x2 a)
{
  const long long i = __builtin_bit_cast(long long, a);
  return __builtin_bit_cast(f32x2, (long long)__builtin_amdgcn_mov_dpp(i, 0x150 + Q, 0xF, 0xF, false));
}
template <int K>
__device__ __forceinline__ float bck(f32x2 a) { return bc32<(K >> 1)>((K & 1) ? a.y : a.x); }

// ---- form 0 ----
template <int K>
__device__ __forceinline__ void step0(f32x2 &z, float &v, const f32x2 *w, f32x2 a)
{
  const float vn = bck<(K + 1 < 32 ? K + 1 : 31)>(a);
  z = __builtin_elementwise_fma(w[K], f32x2{v, v}, z);
  __builtin_amdgcn_sched_barrier(0);
  v = vn;
}
__device__ __forceinline__ f32x2 dot0(const f32x2 *w, f32x2 a)
{
  f32x2 z = {0.0f, 0.0f};
  float v = bck<0>(a);
  __builtin_amdgcn_sched_barrier(0);
#define R4(K) step0<K>(z, v, w, a); step0<K + 1>(z, v, w, a); step0<K + 2>(z, v, w, a); step0<K + 3>(z, v, w, a);
  R4(0) R4(4) R4(8) R4(12) R4(16) R4(20) R4(24) R4(28)
#undef R4
  return z;
}
// ---- form 1 ----
template <int K>
__device__ __forceinline__ void step1(f32x2 &ze, f32x2 &zo, float &v0, float &v1, const f32x2 *w, f32x2 a)
{
  const float n0 = bck<(K + 2 < 32 ? K + 2 : 30)>(a);
  ze = __builtin_elementwise_fma(w[K], f32x2{v0, v0}, ze);
  __builtin_amdgcn_sched_barrier(0);
  const float n1 = bck<(K + 3 < 32 ? K + 3 : 31)>(a);
  zo = __builtin_elementwise_fma(w[K + 1], f32x2{v1, v1}, zo);
  __builtin_amdgcn_sched_barrier(0);
  v0 = n0; v1 = n1;
}
__device__ __forceinline__ f32x2 dot1(const f32x2 *w, f32x2 a)
{
  f32x2 ze = {0.0f, 0.0f}, zo = {0.0f, 0.0f};
  float v0 = bck<0>(a), v1 = bck<1>(a);
  __builtin_amdgcn_sched_barrier(0);
#define R4(K) step1<K>(ze, zo, v0, v1, w, a); step1<K + 2>(ze, zo, v0, v1, w, a);
  R4(0) R4(4) R4(8) R4(12) R4(16) R4(20) R4(24) R4(28)
#undef R4
  return ze + zo;
}
// ---- form 2 ----
template <int Q>
__device__ __forceinline__ void step2(f32x2 &ze, f32x2 &zo, f32x2 &b, const f32x2 *w, f32x2 a)
{
  const f32x2 bn = bc64<(Q + 1 < 16 ? Q + 1 : 15)>(a);
  ze = __builtin_elementwise_fma(w[2 * Q], f32x2{b.x, b.x}, ze);
  zo = __builtin_elementwise_fma(w[2 * Q + 1], f32x2{b.y, b.y}, zo);
  __builtin_amdgcn_sched_barrier(0);
  b = bn;
}
__device__ __forceinline__ f32x2 dot2(const f32x2 *w, f32x2 a)
{
  f32x2 ze = {0.0f, 0.0f}, zo = {0.0f, 0.0f};
  f32x2 b = bc64<0>(a);
  __builtin_amdgcn_sched_barrier(0);
#define R4(Q) step2<Q>(ze, zo, b, w, a); step2<Q + 1>(ze, zo, b, w, a); step2<Q + 2>(ze, zo, b, w, a); step2<Q + 3>(ze, zo, b, w, a);
  R4(0) R4(4) R4(8) R4(12)
#undef R4
  return ze + zo;
}
// ---- form 3 ----
template <int Q>
__device__ __forceinline__ void step3(f32x2 *z, f32x2 &b0, f32x2 &b1, const f32x2 *w, f32x2 a)
{
  const f32x2 n0 = bc64<(Q + 2 < 16 ? Q + 2 : 14)>(a);
  z[0] = __builtin_elementwise_fma(w[2 * Q], f32x2{b0.x, b0.x}, z[0]);
  z[1] = __builtin_elementwise_fma(w[2 * Q + 1], f32x2{b0.y, b0.y}, z[1]);
  __builtin_amdgcn_sched_barrier(0);
  const f32x2 n1 = bc64<(Q + 3 < 16 ? Q + 3 : 15)>(a);
  z[2] = __builtin_elementwise_fma(w[2 * Q + 2], f32x2{b1.x, b1.x}, z[2]);
  z[3] = __builtin_elementwise_fma(w[2 * Q + 3], f32x2{b1.y, b1.y}, z[3]);
  __builtin_amdgcn_sched_barrier(0);
  b0 = n0; b1 = n1;
}
__device__ __forceinline__ f32x2 dot3(const f32x2 *w, f32x2 a)
{
  f32x2 z[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
  f32x2 b0 = bc64<0>(a), b1 = bc64<1>(a);
  __builtin_amdgcn_sched_barrier(0);
#define R4(Q) step3<Q>(z, b0, b1, w, a); step3<Q + 2>(z, b0, b1, w, a);
  R4(0) R4(4) R4(8) R4(12)
#undef R4
  return (z[0] + z[1]) + (z[2] + z[3]);
}

// ---- form 4 / 5 ----
template <int Q, bool MID>
__device__ __forceinline__ void step4(f32x2 &z, f32x2 &b, const f32x2 *w, f32x2 a)
{
  if constexpr (MID) {
    z = __builtin_elementwise_fma(w[2 * Q], f32x2{b.x, b.x}, z);
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 bn = bc64<(Q + 1 < 16 ? Q + 1 : 15)>(a);
    __builtin_amdgcn_sched_barrier(0);
    z = __builtin_elementwise_fma(w[2 * Q + 1], f32x2{b.y, b.y}, z);
    __builtin_amdgcn_sched_barrier(0);
    b = bn;
  } else {
    const f32x2 bn = bc64<(Q + 1 < 16 ? Q + 1 : 15)>(a);
    z = __builtin_elementwise_fma(w[2 * Q], f32x2{b.x, b.x}, z);
    z = __builtin_elementwise_fma(w[2 * Q + 1], f32x2{b.y, b.y}, z);
    __builtin_amdgcn_sched_barrier(0);
    b = bn;
  }
}
template <bool MID>
__device__ __forceinline__ f32x2 dot4(const f32x2 *w, f32x2 a)
{
  f32x2 z = {0.0f, 0.0f};
  f32x2 b = bc64<0>(a);
  __builtin_amdgcn_sched_barrier(0);
#define R4(Q) step4<Q, MID>(z, b, w, a); step4<Q + 1, MID>(z, b, w, a); step4<Q + 2, MID>(z, b, w, a); step4<Q + 3, MID>(z, b, w, a);
  R4(0) R4(4) R4(8) R4(12)
#undef R4
  return z;
}

// ---- forms 6 / 7: the tanh of the layer (csrc/mppi_device.hpp: tanh_bias2) ----
__device__ __forceinline__ f32x2 tanh2(f32x2 z, f32x2 bs)
{
  const f32x2 y = __builtin_elementwise_fma(z, f32x2{2.88539008f, 2.88539008f}, bs);
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(y.x);
  e.y = __builtin_amdgcn_exp2f(y.y);
  const f32x2 d = e + f32x2{1.0f, 1.0f};
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r, f32x2{1.0f, 1.0f});
}
// form 7: one k-half (16 links) per lane; w[j] = the lane's two neurons x input 16 half + j
template <int I>
__device__ __forceinline__ void step7(f32x2 &z, f32x2 &b, const f32x2 *w, f32x2 a)
{
  const f32x2 bn = bc64<(I + 1 < 8 ? I + 1 : 7)>(a);
  z = __builtin_elementwise_fma(w[2 * I], f32x2{b.x, b.x}, z);
  z = __builtin_elementwise_fma(w[2 * I + 1], f32x2{b.y, b.y}, z);
  __builtin_amdgcn_sched_barrier(0);
  b = bn;
}
__device__ __forceinline__ f32x2 dot7(const f32x2 *w, f32x2 a)
{
  f32x2 z = {0.0f, 0.0f};
  f32x2 b = bc64<0>(a);
  __builtin_amdgcn_sched_barrier(0);
  step7<0>(z, b, w, a); step7<1>(z, b, w, a); step7<2>(z, b, w, a); step7<3>(z, b, w, a);
  step7<4>(z, b, w, a); step7<5>(z, b, w, a); step7<6>(z, b, w, a); step7<7>(z, b, w, a);
  // the other half of k sits in the other DPP row of the rollout: even row + odd row, the same order in both rows
  const auto sx = __builtin_amdgcn_permlane16_swap(__float_as_uint(z.x), __float_as_uint(z.x), false, false);
  const auto sy = __builtin_amdgcn_permlane16_swap(__float_as_uint(z.y), __float_as_uint(z.y), false, false);
  return f32x2{__uint_as_float(sx[0]), __uint_as_float(sy[0])} + f32x2{__uint_as_float(sx[1]), __uint_as_float(sy[1])};
}
// rows 1 and 3 (the upper k-half) hold pair (p + 8) & 15 in lane p: row_ror:8 on those rows only
__device__ __forceinline__ f32x2 rot8_upper(f32x2 a)
{
  const long long i = __builtin_bit_cast(long long, a);
  return __builtin_bit_cast(f32x2, (long long)__builtin_amdgcn_update_dpp(i, i, 0x128, 0xA, 0xF, false));
}

__global__ __launch_bounds__(512) void ub7_kernel(const float *wsrc, float *out, long long *ticks, int iters)
{
  const int lane = threadIdx.x & 63, p = lane & 15, half = (lane >> 4) & 1;
  f32x2 w[16];
#pragma unroll
  for (int j = 0; j < 16; j++) w[j] = f32x2{wsrc[(2 * p) * H + 16 * half + j], wsrc[(2 * p + 1) * H + 16 * half + j]};
#pragma unroll
  for (int j = 0; j < 16; j++) asm volatile("" : "+v"(w[j]));
  const int pp = half ? ((p + 8) & 15) : p;  // the pair this lane's activation register holds
  const f32x2 a_init = {0.01f * (float)(2 * pp) + 0.003f, 0.01f * (float)(2 * pp + 1) - 0.002f};
  const f32x2 b_own = {0.01f * (float)(2 * p) + 0.003f, 0.01f * (float)(2 * p + 1) - 0.002f};
  f32x2 a = a_init;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int i = 0; i < iters; i++) {
    const f32x2 z = dot7(w, a);            // lane p of BOTH rows: the sums of pair p
    a = rot8_upper(tanh2(z, b_own));       // the upper row turns them round for the next layer's broadcasts
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  if (lane == 0) { ticks[2 * (threadIdx.x >> 6)] = c1 - c0; ticks[2 * (threadIdx.x >> 6) + 1] = t1 - t0; }
  if (threadIdx.x < 64) { out[threadIdx.x * 2] = a.x; out[threadIdx.x * 2 + 1] = a.y; }
}

template <int FORM>
__global__ __launch_bounds__(256) void ub_kernel(const float *wsrc, float *out, long long *ticks, int iters)
{
  const int lane = threadIdx.x & 63, p = lane & 15;
  f32x2 w[H];
#pragma unroll
  for (int k = 0; k < H; k++) w[k] = f32x2{wsrc[(2 * p) * H + k], wsrc[(2 * p + 1) * H + k]};
#pragma unroll
  for (int k = 0; k < H; k++) asm volatile("" : "+v"(w[k]));
  const f32x2 a_init = {0.01f * (float)(2 * p) + 0.003f, 0.01f * (float)(2 * p + 1) - 0.002f};
  f32x2 a = a_init;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int i = 0; i < iters; i++) {
    f32x2 z;
    if (FORM == 0) z = dot0(w, a);
    else if (FORM == 1) z = dot1(w, a);
    else if (FORM == 2) z = dot2(w, a);
    else if (FORM == 3) z = dot3(w, a);
    else if (FORM == 4 || FORM == 6) z = dot4<false>(w, a);
    else z = dot4<true>(w, a);
    if (FORM == 6) a = tanh2(z, a_init);
    else a = z + a_init;  // (the values stay O(0.1): the bit comparisons below mean something)
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  if (lane == 0) { ticks[2 * (threadIdx.x >> 6)] = c1 - c0; ticks[2 * (threadIdx.x >> 6) + 1] = t1 - t0; }
  out[threadIdx.x * 2] = a.x; out[threadIdx.x * 2 + 1] = a.y;
}

int main()
{
  std::vector<float> hw(H * H);
  unsigned s = 12345;
  for (auto &x : hw) { s = s * 1664525u + 1013904223u; x = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.34f; }
  float *dw, *dout; long long *dt;
  hipMalloc(&dw, hw.size() * 4); hipMalloc(&dout, 512 * 4); hipMalloc(&dt, 8 * 8);
  hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  const int iters = 2000;
  float ref[7][4] = {};
  for (int waves = 1; waves <= 4; waves *= 4) {  // one wave (one SIMD) / four waves (one per SIMD)
    for (int form = 0; form < 7; form++) {
      for (int rep = 0; rep < 3; rep++) {
        switch (form) {
          case 0: hipLaunchKernelGGL(ub_kernel<0>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 1: hipLaunchKernelGGL(ub_kernel<1>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 2: hipLaunchKernelGGL(ub_kernel<2>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 3: hipLaunchKernelGGL(ub_kernel<3>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 4: hipLaunchKernelGGL(ub_kernel<4>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 5: hipLaunchKernelGGL(ub_kernel<5>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          default: hipLaunchKernelGGL(ub_kernel<6>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
        }
        hipDeviceSynchronize();
      }
      long long ht[8]; float ho[8];
      hipMemcpy(ht, dt, sizeof(ht), hipMemcpyDeviceToHost);
      hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
      for (int i = 0; i < 4; i++) ref[form][i] = ho[i];
      printf("waves %d form %d: %.1f shader cycles per layer (clock64), %.2f ns (wall_clock64 at 100 MHz)   out %.9g %.9g\n", waves, form,
             (double)ht[0] / iters, (double)ht[1] * 10.0 / iters, ho[0], ho[1]);
    }
  }
  // form 7: eight waves, two per SIMD (a workgroup's waves go to the SIMDs in turn: waves w and w + 4 share one)
  {
    long long *dt8; hipMalloc(&dt8, 16 * 8);
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(ub7_kernel, dim3(1), dim3(512), 0, 0, dw, dout, dt8, iters); hipDeviceSynchronize(); }
    long long ht[16]; float ho[8];
    hipMemcpy(ht, dt8, sizeof(ht), hipMemcpyDeviceToHost);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1ll << 62;
    for (int w = 0; w < 8; w++) { mx = ht[2 * w] > mx ? ht[2 * w] : mx; mn = ht[2 * w] < mn ? ht[2 * w] : mn; }
    printf("waves 8 form 7 (two per SIMD, k halves, + tanh): %.1f .. %.1f shader cycles per layer over the eight waves (form 6, one wave per SIMD, the same 4 rollouts per SIMD: above)   out %.9g %.9g (form 6: %.9g %.9g)\n",
           (double)mn / iters, (double)mx / iters, ho[0], ho[1], ref[6][0], ref[6][1]);
  }
  printf("forms 2 and 1 agree: %s; forms 4, 5 and 0 agree: %s; forms 0 and 2 differ by %.3g (re-association)\n",
         (ref[1][0] == ref[2][0] && ref[1][1] == ref[2][1]) ? "bit for bit" : "NO",
         (ref[4][0] == ref[0][0] && ref[4][1] == ref[0][1] && ref[5][0] == ref[0][0] && ref[5][1] == ref[0][1]) ? "bit for bit" : "NO",
         (double)(ref[0][0] - ref[2][0]));
  return 0;
}
