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
    else if (FORM == 4) z = dot4<false>(w, a);
    else z = dot4<true>(w, a);
    a = z + a_init;  // (the values stay O(0.1): the bit comparisons below mean something)
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
  float ref[6][4] = {};
  for (int waves = 1; waves <= 4; waves *= 4) {  // one wave (one SIMD) / four waves (one per SIMD)
    for (int form = 0; form < 6; form++) {
      for (int rep = 0; rep < 3; rep++) {
        switch (form) {
          case 0: hipLaunchKernelGGL(ub_kernel<0>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 1: hipLaunchKernelGGL(ub_kernel<1>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 2: hipLaunchKernelGGL(ub_kernel<2>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 3: hipLaunchKernelGGL(ub_kernel<3>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          case 4: hipLaunchKernelGGL(ub_kernel<4>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
          default: hipLaunchKernelGGL(ub_kernel<5>, dim3(1), dim3(64 * waves), 0, 0, dw, dout, dt, iters); break;
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
  printf("forms 2 and 1 agree: %s; forms 4, 5 and 0 agree: %s; forms 0 and 2 differ by %.3g (re-association)\n",
         (ref[1][0] == ref[2][0] && ref[1][1] == ref[2][1]) ? "bit for bit" : "NO",
         (ref[4][0] == ref[0][0] && ref[4][1] == ref[0][1] && ref[5][0] == ref[0][0] && ref[5][1] == ref[0][1]) ? "bit for bit" : "NO",
         (double)(ref[0][0] - ref[2][0]));
  return 0;
}
