// row_lds_ub.hip -- the network recurrence in a VECTOR-ALU "row" form built for LATENCY, measured alone on a SIMD.
// A dependent k-step of v_mfma_f32_16x16x4_f32 costs ~8 cycles (4 k per 32-cycle instruction); a dependent
// v_pk_fma_f32 costs 4 and carries two neurons.  So: one wave = 4 rollouts x 16 lanes, a lane owns neurons (2p, 2p+1) of
// its rollout, weights in registers as pairs, every dot product the k-ascending fmaf chain (bit-identical to the other
// forms); the activations of a layer go through LDS: one 8-B write per lane, eight 16-B broadcast reads per lane.
// Printed: cycles per step (s_memtime) for 6-32-32-4, one and two waves per SIMD, and a checksum against the host.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ub/row_lds_ub.hip -o row_lds_ub
#include "../../autorally_amd/csrc/mppi_device.hpp"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace mppi;

constexpr int H = 32;

struct RowWeights {
  f32x2 w1[6], w2[H], w3[H];
  f32x2 b1s, b2s, b3;  // hidden biases pre-scaled for tanh_bias2
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
  float st[4][4];       // s3..s6 per rollout of the wave
  float ctl[4][4];      // u0, u1 (clamped) per rollout
  float act[2][4][H];   // activations of layer 0 / layer 1
};

// one step: returns this lane's pair of the output layer (d0,d1) for p even, (d2,d3) for p odd (before the bias)
__device__ __forceinline__ f32x2 row_step(const RowWeights &W, RowLds &L, int r, int p)
{
  // layer 0: inputs [s3, s4, s5, s6, u0, u1] of rollout r, broadcast reads
  const float4 s = *reinterpret_cast<const float4 *>(&L.st[r][0]);
  const float2 u = *reinterpret_cast<const float2 *>(&L.ctl[r][0]);
  f32x2 z = {0.0f, 0.0f};
  z = __builtin_elementwise_fma(W.w1[0], f32x2{s.x, s.x}, z);
  z = __builtin_elementwise_fma(W.w1[1], f32x2{s.y, s.y}, z);
  z = __builtin_elementwise_fma(W.w1[2], f32x2{s.z, s.z}, z);
  z = __builtin_elementwise_fma(W.w1[3], f32x2{s.w, s.w}, z);
  z = __builtin_elementwise_fma(W.w1[4], f32x2{u.x, u.x}, z);
  z = __builtin_elementwise_fma(W.w1[5], f32x2{u.y, u.y}, z);
  f32x2 a = tanh_bias2(z, W.b1s);
  *reinterpret_cast<f32x2 *>(&L.act[0][r][2 * p]) = a;
  __builtin_amdgcn_wave_barrier();
  // layer 1
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
    a = tanh_bias2(z, W.b2s);
    *reinterpret_cast<f32x2 *>(&L.act[1][r][2 * p]) = a;
  }
  __builtin_amdgcn_wave_barrier();
  // output layer (every lane runs the chain; lanes p and p+2, p+4, .. hold duplicates)
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
  }
  return z;
}

__global__ __launch_bounds__(256) void k_row(const float *theta, float *out, unsigned long long *cyc, int iters, float dt)
{
  __shared__ __attribute__((aligned(16))) RowLds lds[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane >> 4, p = lane & 15;
  RowWeights W;
  row_load(theta, p, W);
  RowLds &L = lds[w];
  // this lane's pair of the state: (s3, s4) for p even, (s5, s6) for p odd
  const int gk = (blockIdx.x * 4 + w) * 4 + r;  // global rollout
  f32x2 sp = (p & 1) ? f32x2{0.1f, 0.0f} : f32x2{0.01f * (float)(gk % 64), 5.0f};
  *reinterpret_cast<f32x2 *>(&L.st[r][2 * (p & 1)]) = sp;
  *reinterpret_cast<float2 *>(&L.ctl[r][0]) = make_float2(0.1f, 0.1f);
  __builtin_amdgcn_wave_barrier();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    const f32x2 z = row_step(W, L, r, p);
    const f32x2 d = z + W.b3;
    sp = __builtin_elementwise_fma(d, f32x2{dt, dt}, sp);  // incrementState
    *reinterpret_cast<f32x2 *>(&L.st[r][2 * (p & 1)]) = sp;
    __builtin_amdgcn_wave_barrier();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + w] = c1 - c0;
  if (p < 2) { out[gk * 4 + 2 * p] = sp.x; out[gk * 4 + 2 * p + 1] = sp.y; }
}

// The product's form of the step (rollout_row.hip): the new state reaches layer 0 through one DPP move (every lane pair
// computes the same two outputs) instead of a third LDS round trip.  With s_memtime stamps around the three layers.
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <bool SC>
__device__ __forceinline__ f32x2 tanh2(f32x2 z, f32x2 b)
{
  if (SC) return f32x2{tanh_bias(z.x, b.x), tanh_bias(z.y, b.y)};
  return tanh_bias2(z, b);
}
template <bool STAMPS, bool SC = false>
__global__ __launch_bounds__(256) void k_row_dpp(const float *theta, float *out, unsigned long long *cyc, int iters, float dt)
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
  unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, acc[3] = {0, 0, 0};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (STAMPS) STAMP(q0);
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
    *reinterpret_cast<f32x2 *>(&L.act[0][r][2 * p]) = tanh2<SC>(z, W.b1s);
    __builtin_amdgcn_wave_barrier();
    if (STAMPS) STAMP(q1);
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
      *reinterpret_cast<f32x2 *>(&L.act[1][r][2 * p]) = tanh2<SC>(z, W.b2s);
    }
    __builtin_amdgcn_wave_barrier();
    if (STAMPS) STAMP(q2);
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
    if (STAMPS) { STAMP(q3); acc[0] += q1 - q0; acc[1] += q2 - q1; acc[2] += q3 - q2; }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + w] = c1 - c0;
  if (STAMPS && blockIdx.x == 0 && w == 0 && lane == 0) { cyc[4096] = acc[0]; cyc[4097] = acc[1]; cyc[4098] = acc[2]; }
  if (p < 2) { out[gk * 4 + 2 * p] = sp.x; out[gk * 4 + 2 * p + 1] = sp.y; }
}

static float tanh_host(float z, float b)  // the device formula cannot be reproduced bit-exactly on the host: tolerance
{
  return tanhf(z + b);
}

int main(int argc, char **argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 100;
  std::vector<float> th(H * 6 + H + H * H + H + 4 * H + 4);
  for (size_t i = 0; i < th.size(); i++) th[i] = 0.25f * (float)((int)((i * 2654435761u) >> 20 & 255) - 128) / 128.0f;
  float *d_t, *d_o; unsigned long long *d_c;
  hipMalloc(&d_t, th.size() * 4); hipMemcpy(d_t, th.data(), th.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_o, 1024 * 16 * 4 * 4); hipMalloc(&d_c, 4100 * 8);
  const float dt = 0.02f;
  for (int blocks : {256, 512, 1024}) {
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_row, dim3(blocks), dim3(256), 0, 0, d_t, d_o, d_c, iters, dt);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks * 4);
    hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; unsigned long long mx = 0;
    for (auto v : c) { s += (double)v; mx = std::max(mx, v); }
    printf("6-32-32-4 row/LDS form, %d wave(s) per SIMD: %.0f cycles per step per wave (mean), %.0f (slowest wave)\n", blocks / 256,
           s / c.size() / iters, (double)mx / iters);
  }
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k_row_dpp<false, true>), dim3(256), dim3(256), 0, 0, d_t, d_o, d_c, iters, dt);
  hipDeviceSynchronize();
  {
    std::vector<unsigned long long> c(1024);
    hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
    double s2 = 0;
    for (auto v : c) s2 += (double)v;
    printf("DPP form with scalar tanh (tanh_bias per value), 1 wave per SIMD: %.0f cycles per step\n", s2 / 1024 / iters);
  }
  for (int st = 0; st < 2; st++) {
    for (int rep = 0; rep < 3; rep++) {
      if (st) hipLaunchKernelGGL(k_row_dpp<true>, dim3(256), dim3(256), 0, 0, d_t, d_o, d_c, iters, dt);
      else hipLaunchKernelGGL(k_row_dpp<false>, dim3(256), dim3(256), 0, 0, d_t, d_o, d_c, iters, dt);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(4099);
    hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; i++) s += (double)c[i];
    printf("DPP form%s, 1 wave per SIMD: %.0f cycles per step", st ? " (stamped)" : "", s / 1024 / iters);
    if (st) printf("; layer 0 (dpp, 6 fma, tanh, write) %.0f | layer 1 (8 reads, 32 fma, tanh, write) %.0f | output layer (8 reads, 32 fma, Euler) %.0f",
                   (double)c[4096] / iters, (double)c[4097] / iters, (double)c[4098] / iters);
    printf("\n");
  }
  // checksum of rollout 0 against a host replay of the same recurrence (libm tanhf: tolerance)
  std::vector<float> o(16); hipMemcpy(o.data(), d_o, 64, hipMemcpyDeviceToHost);
  const float *W1 = th.data(), *B1 = W1 + H * 6, *W2 = B1 + H, *B2 = W2 + H * H, *W3 = B2 + H, *B3 = W3 + 4 * H;
  float s[4] = {0.0f, 5.0f, 0.1f, 0.0f};
  for (int i = 0; i < iters; i++) {
    float in[6] = {s[0], s[1], s[2], s[3], 0.1f, 0.1f}, a[H], b[H];
    for (int j = 0; j < H; j++) { float z = 0; for (int k = 0; k < 6; k++) z = fmaf(W1[j * 6 + k], in[k], z); a[j] = tanh_host(z, B1[j]); }
    for (int j = 0; j < H; j++) { float z = 0; for (int k = 0; k < H; k++) z = fmaf(W2[j * H + k], a[k], z); b[j] = tanh_host(z, B2[j]); }
    for (int o2 = 0; o2 < 4; o2++) { float z = 0; for (int k = 0; k < H; k++) z = fmaf(W3[o2 * H + k], b[k], z); s[o2] = fmaf(z + B3[o2], dt, s[o2]); }
  }
  printf("rollout 0 after %d steps: device [%g %g %g %g], host [%g %g %g %g]\n", iters, o[0], o[1], o[2], o[3], s[0], s[1], s[2], s[3]);
  return 0;
}
