// row64_ub.hip -- can the latency form of rollout_row.hip be carried to 64-WIDE nets?
// One 64 -> 64 hidden layer + tanh as the k-ascending fmaf chain of neural_net_model.cu:379-394, repeated `iters` times
// (the output of a layer is the input of the next), measured per layer.
//
// Layout under test: a rollout is 32 lanes = two 16-lane DPP rows, lane g of the rollout owns neurons 2g, 2g+1 (a packed
// pair), a wave carries two rollouts.  `row_newbcast` reaches only the 16 lanes of a row, so after the tanh every lane
// fetches the pair of the same lane of the OTHER row of its rollout with one v_permlane16_swap_b32 per component (gfx950):
// L = the pair of the rollout's lower row, U = of its upper row, in every lane; activation k then comes from lane
// (k >> 1) & 15 of L (k < 32) or U by one v_mov_b32_dpp, as in the 32-wide form.
// Weights: 64 packed pairs per lane and layer = 128 VGPRs -- one layer fits, the three 64x64 layers of the shipped
// 6-64-64-64-64-4 do not.  Forms:
//   R   weights in registers
//   L   weights from LDS, one ds_read_b64 per k (both rollouts of a wave read the same 256 B: 2 LDS cycles per
//       wave-instruction, MI355X_MICROARCH.md LDS table), requested PF k-steps ahead
// each with one and with two such waves per SIMD (4 / 8 waves per workgroup, one workgroup per CU).
// The check: all forms give the bits of a scalar CPU statement of the chain (same fmaf order; tanh by the device formula
// is compared at 1e-6).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ub/row64_ub.hip -o row64_ub && ./row64_ub
#include "../../autorally_amd/csrc/mppi_device.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace mppi;

constexpr int H = 64;

template <int Q>
__device__ __forceinline__ float bc(float a)
{
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x150 + Q, 0xF, 0xF, false));  // row_newbcast:Q
}
// activation k of this lane's rollout: lane (k >> 1) & 15 of L (k < 32) or U, component k & 1
template <int K>
__device__ __forceinline__ float act_k(f32x2 L, f32x2 U)
{
  constexpr int q = (K >> 1) & 15;
  return bc<q>(K < 32 ? ((K & 1) ? L.y : L.x) : ((K & 1) ? U.y : U.x));
}
// L / U from this lane's own pair: v_permlane16_swap_b32 swaps the odd rows of its first operand with the even rows of its
// second; on two copies of `a` that leaves (lower row's value, lower row's value) in the first and (upper, upper) in the second
__device__ __forceinline__ void swap_rows(f32x2 a, f32x2 &L, f32x2 &U)
{
  auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(a.x), false, false);
  auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(a.y), false, false);
  L = f32x2{__uint_as_float(x[0]), __uint_as_float(y[0])};
  U = f32x2{__uint_as_float(x[1]), __uint_as_float(y[1])};
}

template <int K>
__device__ __forceinline__ void step_r(f32x2 &z, float &v, const f32x2 *w, f32x2 L, f32x2 U)
{
  const float vn = act_k<(K + 1 < H ? K + 1 : H - 1)>(L, U);
  z = __builtin_elementwise_fma(w[K], f32x2{v, v}, z);
  __builtin_amdgcn_sched_barrier(0);
  v = vn;
}
template <int K0>
__device__ __forceinline__ void steps16_r(f32x2 &z, float &v, const f32x2 *w, f32x2 L, f32x2 U)
{
#define S4(K) step_r<K>(z, v, w, L, U); step_r<K + 1>(z, v, w, L, U); step_r<K + 2>(z, v, w, L, U); step_r<K + 3>(z, v, w, L, U);
  S4(K0) S4(K0 + 4) S4(K0 + 8) S4(K0 + 12)
#undef S4
}

// ---- form R: weights in registers ----
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_reg(const float *W, const float *B, float *out, unsigned long long *cyc, int iters)
{
  const int lane = threadIdx.x & 63, g = lane & 31;
  f32x2 w[H];
#pragma unroll
  for (int k = 0; k < H; k++) w[k] = f32x2{W[(2 * g) * H + k], W[(2 * g + 1) * H + k]};
  const f32x2 bs = f32x2{B[2 * g] * kTanhScale, B[2 * g + 1] * kTanhScale};
#pragma unroll
  for (int k = 0; k < H; k++) asm volatile("" : "+v"(w[k]));
  f32x2 a = f32x2{0.01f * (float)(2 * g) - 0.3f, 0.01f * (float)(2 * g + 1) - 0.3f};
  if (lane >= 32) a = a * f32x2{-0.5f, -0.5f};  // the second rollout of the wave starts elsewhere
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    f32x2 L, U;
    swap_rows(a, L, U);
    f32x2 z = {0.0f, 0.0f};
    float v = act_k<0>(L, U);
    __builtin_amdgcn_sched_barrier(0);
    steps16_r<0>(z, v, w, L, U);
    steps16_r<16>(z, v, w, L, U);
    steps16_r<32>(z, v, w, L, U);
    steps16_r<48>(z, v, w, L, U);
    a = tanh_bias2(z, bs);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = c1 - c0;
  if (blockIdx.x == 0 && threadIdx.x < 64) { out[2 * lane] = a.x; out[2 * lane + 1] = a.y; }
}

typedef const volatile f32x2 __attribute__((address_space(3))) *lds_p;
template <int K, int PF>
__device__ __forceinline__ void step_l(f32x2 &z, float &v, f32x2 *ring, lds_p base, f32x2 L, f32x2 U)
{
  const f32x2 wk = ring[K % PF];
  if constexpr (K + PF < H) ring[K % PF] = base[(K + PF) * 32];
  const float vn = act_k<(K + 1 < H ? K + 1 : H - 1)>(L, U);
  z = __builtin_elementwise_fma(wk, f32x2{v, v}, z);
  __builtin_amdgcn_sched_barrier(0);
  v = vn;
}
template <int K0, int PF>
__device__ __forceinline__ void steps16_l(f32x2 &z, float &v, f32x2 *ring, lds_p base, f32x2 L, f32x2 U)
{
#define S4(K) step_l<K, PF>(z, v, ring, base, L, U); step_l<K + 1, PF>(z, v, ring, base, L, U); step_l<K + 2, PF>(z, v, ring, base, L, U); step_l<K + 3, PF>(z, v, ring, base, L, U);
  S4(K0) S4(K0 + 4) S4(K0 + 8) S4(K0 + 12)
#undef S4
}

// ---- form L: weights from LDS, PF k-steps ahead ----
// image: wl[k][g] = (W[2g][k], W[2g+1][k]) -- a wave's read of step k is 32 consecutive 8-B entries, twice
template <int WAVES, int PF>
__global__ __launch_bounds__(64 * WAVES) void k_lds(const float *W, const float *B, float *out, unsigned long long *cyc, int iters)
{
  __shared__ __attribute__((aligned(16))) f32x2 wl[H][32];
  const int lane = threadIdx.x & 63, g = lane & 31;
  for (int i = threadIdx.x; i < H * 32; i += 64 * WAVES) {
    const int k = i >> 5, gg = i & 31;
    wl[k][gg] = f32x2{W[(2 * gg) * H + k], W[(2 * gg + 1) * H + k]};
  }
  __syncthreads();
  const f32x2 bs = f32x2{B[2 * g] * kTanhScale, B[2 * g + 1] * kTanhScale};
  f32x2 a = f32x2{0.01f * (float)(2 * g) - 0.3f, 0.01f * (float)(2 * g + 1) - 0.3f};
  if (lane >= 32) a = a * f32x2{-0.5f, -0.5f};
  const lds_p base = (lds_p)&wl[0][g];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    f32x2 ring[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) ring[k] = base[k * 32];
    f32x2 L, U;
    swap_rows(a, L, U);
    f32x2 z = {0.0f, 0.0f};
    float v = act_k<0>(L, U);
    __builtin_amdgcn_sched_barrier(0);
    steps16_l<0, PF>(z, v, ring, base, L, U);
    steps16_l<16, PF>(z, v, ring, base, L, U);
    steps16_l<32, PF>(z, v, ring, base, L, U);
    steps16_l<48, PF>(z, v, ring, base, L, U);
    a = tanh_bias2(z, bs);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = c1 - c0;
  if (blockIdx.x == 0 && threadIdx.x < 64) { out[2 * lane] = a.x; out[2 * lane + 1] = a.y; }
}

static float tanh_dev(float z, float b)
{  // tanh_bias2 of mppi_device.hpp on the host (exp2f / division in place of v_exp_f32 / v_rcp_f32: compared at 1e-6)
  const float y = fmaf(z, kTanhScale, b * kTanhScale);
  const float e = exp2f(y);
  return fmaf(-2.0f, 1.0f / (e + 1.0f), 1.0f);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class F>
static int run(const char *name, F launch, int waves, const std::vector<float> &ref, float *d_out, unsigned long long *d_cyc, int blocks, int iters)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> o(128);
  std::vector<unsigned long long> c((size_t)blocks * waves);
  CK(hipMemcpy(o.data(), d_out, 128 * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
  double mx = 0, worst = 0;
  for (int i = 0; i < 128; i++) worst = fmax(worst, fabs((double)o[i] - (double)ref[i]));
  for (auto v : c) mx = fmax(mx, (double)v);
  printf("%-34s %8.1f ns / layer by events, %7.1f s_memtime ticks / layer (slowest wave)   max |out - host| %.2e\n", name, 1e6 * ms / iters,
         mx / iters, worst);
  return 0;
}

int main()
{
  const int iters = 2000, blocks = 256;
  std::vector<float> W(H * H), B(H);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto &x : W) x = rnd() * 0.5f;
  for (auto &x : B) x = rnd() * 0.2f;
  // host statement: two rollouts (lanes 0-31, 32-63), `iters` layers
  std::vector<float> ref(128);
  for (int r = 0; r < 2; r++) {
    float a[H], n[H];
    for (int j = 0; j < H; j++) a[j] = (0.01f * (float)j - 0.3f) * (r ? -0.5f : 1.0f);
    for (int it = 0; it < iters; it++) {
      for (int j = 0; j < H; j++) {
        float z = 0.0f;
        for (int k = 0; k < H; k++) z = fmaf(W[j * H + k], a[k], z);
        n[j] = tanh_dev(z, B[j]);
      }
      memcpy(a, n, sizeof(a));
    }
    for (int j = 0; j < H; j++) ref[r * 64 + j] = a[j];
  }
  float *dW, *dB, *d_out;
  unsigned long long *d_cyc;
  CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&d_out, 128 * 4));
  CK(hipMalloc(&d_cyc, (size_t)blocks * 8 * 8));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  printf("one 64x64 layer + tanh of the 32-lane-rollout row form, %d workgroups, %d layers per launch\n", blocks, iters);
#define RUN(name, kern, waves) if (run(name, [&]() { hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), 0, 0, dW, dB, d_out, d_cyc, iters); }, waves, ref, d_out, d_cyc, blocks, iters)) return 1;
  RUN("R  registers, 1 wave / SIMD", (k_reg<4>), 4)
  RUN("R  registers, 2 waves / SIMD", (k_reg<8>), 8)
  RUN("L  LDS PF=4,  1 wave / SIMD", (k_lds<4, 4>), 4)
  RUN("L  LDS PF=8,  1 wave / SIMD", (k_lds<4, 8>), 4)
  RUN("L  LDS PF=16, 1 wave / SIMD", (k_lds<4, 16>), 4)
  RUN("L  LDS PF=8,  2 waves / SIMD", (k_lds<8, 8>), 8)
  RUN("L  LDS PF=16, 2 waves / SIMD", (k_lds<8, 16>), 8)
  return 0;
}
