// dyn_step_ub.hip -- cycles per step of one dynamics wavefront's network recurrence (mfma_net.hpp), by parts.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form tools/ub/dyn_step_ub.hip -o dyn_step_ub
//   dyn_step_ub [iters]
// Variants (template V): 0 the step as the kernels compile it (packed tanh); 1 tanh with scalar multiply-adds;
// 2 no tanh (bias add only); 3 tanh only (no MFMA, the activations feed the next tanh); 4 MFMA chain only,
// each layer fed by the previous accumulators untouched.  Each at 1, 2 and 4 wavefronts per SIMD
// (blocks of 256 threads = one wave per SIMD of a CU; 256 / 512 / 1024 blocks).
#include "../../autorally_amd/csrc/mfma_net.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace mppi;

template <int H, int NHID, int V>
__device__ __forceinline__ void step(const float (&A)[MfmaNet<H, NHID>::nA], const float (&Bi)[MfmaNet<H, NHID>::nBias],
                                     float b0, float b1, float (&d)[4])
{
  using N = MfmaNet<H, NHID>;
  constexpr int MT = N::MT, KSH = N::KSH;
  f32x4 acc[MT];
  if (V != 3) nn_layer0_ops<H, NHID>(A, b0, b1, acc);
  else
    for (int m = 0; m < MT; m++) acc[m] = f32x4{b0, b1, b0 + b1, b0 - b1};
#pragma unroll
  for (int l = 1; l <= NHID; l++) {
    const int boff = (l - 1) * MT * 4;
    float act[MT * 4];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        if (V == 0 || V == 3) {
          const f32x2 v = tanh_bias2(f32x2{acc[m][r], acc[m][r + 1]}, f32x2{Bi[boff + m * 4 + r], Bi[boff + m * 4 + r + 1]});
          act[m * 4 + r] = v.x; act[m * 4 + r + 1] = v.y;
        } else if (V == 1) {
          act[m * 4 + r] = tanh_bias(acc[m][r], Bi[boff + m * 4 + r]);
          act[m * 4 + r + 1] = tanh_bias(acc[m][r + 1], Bi[boff + m * 4 + r + 1]);
        } else if (V == 2) {
          act[m * 4 + r] = acc[m][r] + Bi[boff + m * 4 + r];
          act[m * 4 + r + 1] = acc[m][r + 1] + Bi[boff + m * 4 + r + 1];
        } else {
          act[m * 4 + r] = acc[m][r]; act[m * 4 + r + 1] = acc[m][r + 1];
        }
      }
    if (l < NHID) {
      const int aoff = N::nA0 + (l - 1) * N::nAH;
      if (V != 3) {
#pragma unroll
        for (int m = 0; m < MT; m++) acc[m] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KSH; s++)
#pragma unroll
          for (int m = 0; m < MT; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[aoff + m * KSH + s], act[s], acc[m], 0, 0, 0);
      } else {
#pragma unroll
        for (int m = 0; m < MT; m++) acc[m] = f32x4{act[m * 4], act[m * 4 + 1], act[m * 4 + 2], act[m * 4 + 3]};
      }
    } else {
      const int aoff = N::nA0 + (NHID - 1) * N::nAH, bl = NHID * MT * 4;
      f32x4 o = {0, 0, 0, 0};
      if (V != 3) {
#pragma unroll
        for (int s = 0; s < KSH; s++) o = __builtin_amdgcn_mfma_f32_16x16x4f32(A[aoff + s], act[s], o, 0, 0, 0);
      } else {
        float t = 0;
#pragma unroll
        for (int s = 0; s < MT * 4; s++) t += act[s];
        o = f32x4{t, t, t, t};
      }
#pragma unroll
      for (int r = 0; r < 4; r++) d[r] = o[r] + Bi[bl + r];
    }
  }
}

template <int H, int NHID, int V>
__global__ __launch_bounds__(256) void k_step(const float *wpack, float *out, unsigned long long *cyc, int iters, float dt)
{
  using N = MfmaNet<H, NHID>;
  const int lane = threadIdx.x & 63, g = lane >> 4;
  float A[N::nA], Bi[N::nBias];
  load_weights<H, NHID>(wpack, lane, A, Bi);
  float s3 = 0.01f * lane, s4 = 5.0f, s5 = 0.1f, s6 = 0.0f, b1 = (g < 2) ? 0.1f : 0.0f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
    float d[4];
    step<H, NHID, V>(A, Bi, b0, b1, d);
    s3 = fmaf(d[0], dt, s3); s4 = fmaf(d[1], dt, s4); s5 = fmaf(d[2], dt, s5); s6 = fmaf(d[3], dt, s6);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
  out[blockIdx.x * 256 + threadIdx.x] = s3 + s4 + s5 + s6;
}

template <int H, int NHID, int V>
void run(const char *name, const float *d_w, float *d_o, unsigned long long *d_c, int iters)
{
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL((k_step<H, NHID, V>), dim3(blocks), dim3(256), 0, 0, d_w, d_o, d_c, iters, 0.02f);
    hipLaunchKernelGGL((k_step<H, NHID, V>), dim3(blocks), dim3(256), 0, 0, d_w, d_o, d_c, iters, 0.02f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks * 4);
    hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    printf("H=%d NHID=%d %-34s %d wave(s)/SIMD: %8.1f cycles per step per wave (median; max %.1f) -> %7.1f per SIMD-step\n", H, NHID, name, wps,
           (double)c[c.size() / 2] / iters, (double)c.back() / iters, (double)c[c.size() / 2] / iters / 1.0);
  }
}

int main(int argc, char **argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  float *d_w, *d_o;
  unsigned long long *d_c;
  std::vector<float> w(64 * 400);
  for (size_t i = 0; i < w.size(); i++) w[i] = 0.3f * (float)((int)((i * 2654435761u) >> 20 & 255) - 128) / 128.0f;
  hipMalloc(&d_w, w.size() * 4);
  hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_o, 1024 * 256 * 4);
  hipMalloc(&d_c, 1024 * 4 * 8);
  run<32, 2, 0>("full, packed tanh (as shipped)", d_w, d_o, d_c, iters);
  run<32, 2, 1>("full, scalar tanh", d_w, d_o, d_c, iters);
  run<32, 2, 2>("no tanh (bias add)", d_w, d_o, d_c, iters);
  run<32, 2, 3>("tanh only (packed)", d_w, d_o, d_c, iters);
  run<32, 2, 4>("MFMA chain only", d_w, d_o, d_c, iters);
  run<64, 2, 0>("full, packed tanh (as shipped)", d_w, d_o, d_c, iters);
  run<64, 2, 1>("full, scalar tanh", d_w, d_o, d_c, iters);
  run<64, 2, 2>("no tanh (bias add)", d_w, d_o, d_c, iters);
  run<64, 2, 3>("tanh only (packed)", d_w, d_o, d_c, iters);
  run<64, 2, 4>("MFMA chain only", d_w, d_o, d_c, iters);
  return 0;
}
