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

// The step plus the per-step glue of multi_dynamics (rollout_multi.hip) with no partner waves: state record +
// publication, the three tracked LDS requests pinned by sched_barriers, their use behind the network, the ring
// checks.  G = 1: as in the kernel; G = 2: without the two sched_barriers; G = 3: glue without the requests
// (record + publication only); G = 4: requests through one asm batch read at the END of the step.
template <int H, int NHID, int G>
__global__ __launch_bounds__(256) void k_glue(const float *wpack, float *out, unsigned long long *cyc, int iters, float dt)
{
  using N = MfmaNet<H, NHID>;
  __shared__ float rec[16][64][4];
  __shared__ int pub[4][64], ctl_pub[64], cost_done[64];
  __shared__ float ctl_b1[16][4][64];
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, w = threadIdx.x >> 6;
  float A[N::nA], Bi[N::nBias];
  load_weights<H, NHID>(wpack, lane, A, Bi);
  for (int q = 0; q < 16; q++) ctl_b1[q][w][lane] = (g < 2) ? 0.1f : 0.0f;
  ctl_pub[lane] = 1 << 30; cost_done[lane] = 1 << 30;
  __syncthreads();
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile float __attribute__((address_space(3))) *lds_float_p;
  const lds_int_p p_pub = (lds_int_p)&ctl_pub[0];
  const lds_int_p p_cd = (lds_int_p)&cost_done[0];
  const lds_float_p p_b1 = (lds_float_p)&ctl_b1[0][w][lane];
  const uint32_t a_mypub = lds_addr(&pub[w][lane]);
  const uint32_t a_rec = lds_addr(&rec[0][16 * w + j][g]);
  float s3 = 0.01f * lane, s4 = 5.0f, s5 = 0.1f, s6 = 0.0f;
  float b1_next = *p_b1;
  if (G >= 5) asm volatile("" : "+v"(b1_next));  // the wait for the pre-loop read stays out of the loop
  int cd = 1 << 30, budget = 1 << 30;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < iters; t++) {
    const float b1 = b1_next;
    const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
    if (G != 8) while (cd < t - 16 + 1 && --budget > 0) cd = __builtin_amdgcn_readfirstlane(*p_cd);
    if (G != 7) {
      asm volatile("ds_write_b32 %0, %1" ::"v"(a_rec + (uint32_t)(t & 15) * (64 * 16)), "v"(b0) : "memory");
      lds_publish(a_mypub, t + 1);
    }
    if (G != 8 && t == iters - 1) break;
    const int tn = (t + 1) & 15;
    int cp = 0, cdn = 0;
    float b1n = 0.0f;
    if (G == 1 || G == 2 || G == 5) {
      cp = *p_pub;
      b1n = p_b1[tn * 256];
      cdn = *p_cd;
    }
    if (G == 1 || G == 5) __builtin_amdgcn_sched_barrier(0);
    float d[4];
    step<H, NHID, 0>(A, Bi, b0, b1, d);
    s3 = fmaf(d[0], dt, s3); s4 = fmaf(d[1], dt, s4); s5 = fmaf(d[2], dt, s5); s6 = fmaf(d[3], dt, s6);
    if (G == 1 || G == 5) __builtin_amdgcn_sched_barrier(0);
    if (G == 4) {
      asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(cp), "=&v"(b1n), "=&v"(cdn)
                   : "v"(lds_addr(&ctl_pub[0])), "v"(lds_addr(&ctl_b1[tn][w][lane])), "v"(lds_addr(&cost_done[0]))
                   : "memory");
    }
    if (G != 3 && G < 6) {
      const int want = t + 2;
      cp = __builtin_amdgcn_readfirstlane(cp);
      while (cp < want && --budget > 0) {
        cp = __builtin_amdgcn_readfirstlane(*p_pub);
        b1n = p_b1[tn * 256];
        cdn = *p_cd;
      }
      b1_next = b1n;
      cd = __builtin_amdgcn_readfirstlane(cdn);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
  out[blockIdx.x * 256 + threadIdx.x] = s3 + s4 + s5 + s6 + (float)budget;
}

// The same glue restructured: no ring check and no exit test at the top of the step (the last step is peeled, the
// ring check joins the end-of-step test of the control wave's count: ONE scalar branch per step).
template <int H, int NHID, int G>
__global__ __launch_bounds__(256) void k_glue2(const float *wpack, float *out, unsigned long long *cyc, int iters, float dt)
{
  using N = MfmaNet<H, NHID>;
  __shared__ float rec[16][64][4];
  __shared__ int pub[4][64], ctl_pub[64], cost_done[64];
  __shared__ float ctl_b1[16][4][64];
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, w = threadIdx.x >> 6;
  float A[N::nA], Bi[N::nBias];
  load_weights<H, NHID>(wpack, lane, A, Bi);
  for (int q = 0; q < 16; q++) ctl_b1[q][w][lane] = (g < 2) ? 0.1f : 0.0f;
  ctl_pub[lane] = 1 << 30; cost_done[lane] = 1 << 30;
  __syncthreads();
  typedef const volatile int __attribute__((address_space(3))) *lds_int_p;
  typedef const volatile float __attribute__((address_space(3))) *lds_float_p;
  const lds_int_p p_pub = (lds_int_p)&ctl_pub[0];
  const lds_int_p p_cd = (lds_int_p)&cost_done[0];
  const lds_float_p p_b1 = (lds_float_p)&ctl_b1[0][w][lane];
  const uint32_t a_mypub = lds_addr(&pub[w][lane]);
  const uint32_t a_rec = lds_addr(&rec[0][16 * w + j][g]);
  float s3 = 0.01f * lane, s4 = 5.0f, s5 = 0.1f, s6 = 0.0f;
  float b1 = *p_b1;
  asm volatile("" : "+v"(b1));
  int budget = 1 << 30;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < iters - 1; t++) {
    const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a_rec + (uint32_t)(t & 15) * (64 * 16)), "v"(b0) : "memory");
    lds_publish(a_mypub, t + 1);
    const int tn = (t + 1) & 15;
    int cp = *p_pub;
    float b1n = p_b1[tn * 256];
    int cdn = *p_cd;
    if (G == 1) __builtin_amdgcn_sched_barrier(0);
    float d[4];
    step<H, NHID, 0>(A, Bi, b0, b1, d);
    s3 = fmaf(d[0], dt, s3); s4 = fmaf(d[1], dt, s4); s5 = fmaf(d[2], dt, s5); s6 = fmaf(d[3], dt, s6);
    if (G == 1) __builtin_amdgcn_sched_barrier(0);
    int cps = __builtin_amdgcn_readfirstlane(cp), cds = __builtin_amdgcn_readfirstlane(cdn);
    while (((cps < t + 2) | (cds < t - 14)) && --budget > 0) {
      cps = __builtin_amdgcn_readfirstlane(*p_pub);
      b1n = p_b1[tn * 256];
      cds = __builtin_amdgcn_readfirstlane(*p_cd);
    }
    b1 = b1n;
  }
  {
    const int t = iters - 1;
    const float b0 = (g == 0) ? s3 : (g == 1) ? s4 : (g == 2) ? s5 : s6;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a_rec + (uint32_t)(t & 15) * (64 * 16)), "v"(b0) : "memory");
    lds_publish(a_mypub, t + 1);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
  out[blockIdx.x * 256 + threadIdx.x] = s3 + s4 + s5 + s6 + (float)budget;
}

template <int H, int NHID, int G>
void run_glue2(const char *name, const float *d_w, float *d_o, unsigned long long *d_c, int iters)
{
  const int blocks = 256;
  hipLaunchKernelGGL((k_glue2<H, NHID, G>), dim3(blocks), dim3(256), 0, 0, d_w, d_o, d_c, iters, 0.02f);
  hipLaunchKernelGGL((k_glue2<H, NHID, G>), dim3(blocks), dim3(256), 0, 0, d_w, d_o, d_c, iters, 0.02f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(blocks * 4);
  hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
  std::sort(c.begin(), c.end());
  printf("H=%d NHID=%d %-46s 1 wave/SIMD: %8.1f cycles per step per wave (median; max %.1f)\n", H, NHID, name,
         (double)c[c.size() / 2] / iters, (double)c.back() / iters);
}

template <int H, int NHID, int G>
void run_glue(const char *name, const float *d_w, float *d_o, unsigned long long *d_c, int iters)
{
  const int blocks = 256;
  hipLaunchKernelGGL((k_glue<H, NHID, G>), dim3(blocks), dim3(256), 0, 0, d_w, d_o, d_c, iters, 0.02f);
  hipLaunchKernelGGL((k_glue<H, NHID, G>), dim3(blocks), dim3(256), 0, 0, d_w, d_o, d_c, iters, 0.02f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(blocks * 4);
  hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
  std::sort(c.begin(), c.end());
  printf("H=%d NHID=%d %-46s 1 wave/SIMD: %8.1f cycles per step per wave (median; max %.1f)\n", H, NHID, name,
         (double)c[c.size() / 2] / iters, (double)c.back() / iters);
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
  run_glue<32, 2, 1>("step + glue of multi_dynamics", d_w, d_o, d_c, iters);
  run_glue<32, 2, 2>("step + glue, no sched_barriers", d_w, d_o, d_c, iters);
  run_glue<32, 2, 3>("step + record and publication only", d_w, d_o, d_c, iters);
  run_glue<32, 2, 4>("step + glue, requests in one asm batch at the end", d_w, d_o, d_c, iters);
  run_glue<32, 2, 5>("step + glue, pre-loop read pinned", d_w, d_o, d_c, iters);
  run_glue<32, 2, 6>("step + record and publication, pinned", d_w, d_o, d_c, iters);
  run_glue<32, 2, 7>("step + ring check and exit test only", d_w, d_o, d_c, iters);
  run_glue<32, 2, 8>("step + the two ds_writes only", d_w, d_o, d_c, iters);
  run_glue2<32, 2, 1>("restructured glue (one branch), sched_barriers", d_w, d_o, d_c, iters);
  run_glue2<32, 2, 0>("restructured glue (one branch), no barriers", d_w, d_o, d_c, iters);
  run_glue2<64, 2, 1>("restructured glue (one branch), sched_barriers", d_w, d_o, d_c, iters);
  run_glue2<64, 2, 0>("restructured glue (one branch), no barriers", d_w, d_o, d_c, iters);
  run_glue<64, 2, 1>("step + glue of multi_dynamics", d_w, d_o, d_c, iters);
  run_glue<64, 2, 5>("step + glue, pre-loop read pinned", d_w, d_o, d_c, iters);
  run_glue<64, 2, 3>("step + record and publication only", d_w, d_o, d_c, iters);
  run<64, 2, 0>("full, packed tanh (as shipped)", d_w, d_o, d_c, iters);
  run<64, 2, 1>("full, scalar tanh", d_w, d_o, d_c, iters);
  run<64, 2, 2>("no tanh (bias add)", d_w, d_o, d_c, iters);
  run<64, 2, 3>("tanh only (packed)", d_w, d_o, d_c, iters);
  run<64, 2, 4>("MFMA chain only", d_w, d_o, d_c, iters);
  return 0;
}
