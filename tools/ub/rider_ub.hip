// rider_ub.hip -- how much foreign vector work can ride on the SIMD of a dynamics wavefront for free?
// A workgroup of 5 waves: waves 0..3 run the network recurrence (mfma_net.hpp, one per SIMD of the CU), wave 4 --
// placed on SIMD 0 again by the dispatcher -- is the rider: once per step of wave 0 (LDS sequence word, as in the
// rollout kernels) it executes NR independent v_fma_f32 in chains of 8.  Printed: cycles per step of wave 0 (with
// the rider) and of wave 2 (alone on its SIMD), for several NR, 6-32-32-4 and 6-64-64-4.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form tools/ub/rider_ub.hip -o rider_ub
#include "../../autorally_amd/csrc/mfma_net.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace mppi;

// KIND of the rider's instructions: 0 v_fma_f32, 1 v_exp_f32, 2 v_fma_f64, 3 ds_read_b32 (dependent address chain
// broken: independent reads), 4 global_load_dword (L2 hits), 5 v_rcp_f32
template <int H, int NR, int KIND>
__global__ __launch_bounds__(320) void k_rider(const float *wpack, float *out, unsigned long long *cyc, int iters, float dt)
{
  using N = MfmaNet<H, 2>;
  __shared__ int step_pub[64];
  __shared__ float lbuf[512];
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (role == 0) step_pub[lane] = 0;
  for (int i = threadIdx.x; i < 512; i += 320) lbuf[i] = 0.001f * i;
  __syncthreads();
  if (role < 4) {
    const int g = lane >> 4;
    float A[N::nA], Bi[N::nBias];
    load_weights<H, 2>(wpack, lane, A, Bi);
    float s3 = 0.01f * lane, s4 = 5.0f, s5 = 0.1f, s6 = 0.0f, b1 = (g < 2) ? 0.1f : 0.0f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
      const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
      if (role == 0) lds_publish(lds_addr(&step_pub[lane]), i + 1);
      f32x4 acc[N::MT];
      nn_layer0_ops<H, 2>(A, b0, b1, acc);
      nn_hidden<H, 2>(A, Bi, acc);
      float d[4];
      nn_last<H, 2>(A, Bi, acc, d);
      s3 = fmaf(d[0], dt, s3); s4 = fmaf(d[1], dt, s4); s5 = fmaf(d[2], dt, s5); s6 = fmaf(d[3], dt, s6);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + role] = c1 - c0;
    out[blockIdx.x * 256 + threadIdx.x] = s3 + s4 + s5 + s6;
  } else {
    float f[8];
    double fd[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { f[q] = 0.5f * (q + 1) + lane; fd[q] = 0.25 * (q + 1) + lane; }
    const uint32_t a_pub = lds_addr(&step_pub[0]);
    int seen = 0;
    for (int i = 0; i < iters; i++) {
      int budget = 1 << 20;
      while (seen < i + 1 && --budget > 0) {
        seen = lds_peek(a_pub);
        if (seen < i + 1) __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int n = 0; n < NR / 8; n++)
#pragma unroll
        for (int q = 0; q < 8; q++) {
          if (KIND == 0) f[q] = fmaf(f[q], 1.0001f, 0.5f);
          else if (KIND == 1) f[q] = __builtin_amdgcn_exp2f(f[q]) * 0.5f;
          else if (KIND == 2) fd[q] = fma(fd[q], 1.0001, 0.5);
          else if (KIND == 3) f[q] += ((volatile float *)lbuf)[(lane + 8 * n + q) & 511];
          else if (KIND == 4) f[q] += ((const volatile float *)wpack)[(lane + 64 * (8 * n + q)) & 8191];
          else f[q] = __builtin_amdgcn_rcpf(f[q]) + 1.0f;
        }
    }
    float s = 0;
    for (int q = 0; q < 8; q++) s += f[q] + (float)fd[q];
    if (s == 12345.0f) out[0] = s;
  }
}

template <int H, int NR, int KIND = 0>
void run(const float *d_w, float *d_o, unsigned long long *d_c)
{
  const int iters = 1000, blocks = 256;
  hipLaunchKernelGGL((k_rider<H, NR, KIND>), dim3(blocks), dim3(320), 0, 0, d_w, d_o, d_c, iters, 0.02f);
  hipLaunchKernelGGL((k_rider<H, NR, KIND>), dim3(blocks), dim3(320), 0, 0, d_w, d_o, d_c, iters, 0.02f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(blocks * 4);
  hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> w0, w2;
  for (int b = 0; b < blocks; b++) { w0.push_back((double)c[b * 4] / iters); w2.push_back((double)c[b * 4 + 2] / iters); }
  std::sort(w0.begin(), w0.end()); std::sort(w2.begin(), w2.end());
  static const char *kinds[] = {"v_fma_f32", "v_exp_f32", "v_fma_f64", "ds_read_b32", "global_load", "v_rcp_f32"};
  printf("6-%d-%d-4  rider %4d %-11s per step: dynamics wave WITH the rider %7.1f cycles per step, alone %7.1f  (+%.0f)\n", H, H, NR, kinds[KIND],
         w0[blocks / 2], w2[blocks / 2], w0[blocks / 2] - w2[blocks / 2]);
}

int main()
{
  float *d_w, *d_o; unsigned long long *d_c;
  std::vector<float> w(64 * 400);
  for (size_t i = 0; i < w.size(); i++) w[i] = 0.3f * (float)((int)((i * 2654435761u) >> 20 & 255) - 128) / 128.0f;
  hipMalloc(&d_w, w.size() * 4); hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_o, 256 * 320 * 4); hipMalloc(&d_c, 256 * 4 * 8);
  run<32, 0>(d_w, d_o, d_c); run<32, 32>(d_w, d_o, d_c); run<32, 64>(d_w, d_o, d_c); run<32, 96>(d_w, d_o, d_c);
  run<32, 128>(d_w, d_o, d_c); run<32, 192>(d_w, d_o, d_c); run<32, 256>(d_w, d_o, d_c);
  run<64, 0>(d_w, d_o, d_c); run<64, 32>(d_w, d_o, d_c); run<64, 64>(d_w, d_o, d_c); run<64, 96>(d_w, d_o, d_c);
  run<64, 128>(d_w, d_o, d_c); run<64, 192>(d_w, d_o, d_c); run<64, 256>(d_w, d_o, d_c);
  run<32, 32, 1>(d_w, d_o, d_c); run<32, 64, 1>(d_w, d_o, d_c); run<32, 128, 1>(d_w, d_o, d_c);
  run<32, 32, 5>(d_w, d_o, d_c); run<32, 64, 5>(d_w, d_o, d_c);
  run<32, 32, 2>(d_w, d_o, d_c); run<32, 64, 2>(d_w, d_o, d_c); run<32, 128, 2>(d_w, d_o, d_c);
  run<32, 32, 3>(d_w, d_o, d_c); run<32, 64, 3>(d_w, d_o, d_c);
  run<32, 32, 4>(d_w, d_o, d_c); run<32, 64, 4>(d_w, d_o, d_c);
  run<64, 64, 1>(d_w, d_o, d_c); run<64, 64, 2>(d_w, d_o, d_c); run<64, 64, 3>(d_w, d_o, d_c); run<64, 64, 4>(d_w, d_o, d_c);
  return 0;
}
