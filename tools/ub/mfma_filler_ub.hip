// mfma_filler_ub.hip -- does vector work issue in the shadow of v_mfma_f32_16x16x4_f32 (same wave)?
// A loop of MFMAs on four independent accumulators with NF independent filler instructions after each MFMA;
// prints cycles per MFMA for each filler kind and count.  One wave per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NF, bool DEP>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
  f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
  float a = seed + threadIdx.x * 1e-3f, b = 1.0f - seed;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; i++) f[i] = seed * (i + 1);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int ai = DEP ? 0 : (q & 3);
      acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[ai], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NF; n++) {
        float &x = f[(q * NF + n) & 7];
        if (KIND == 0) x = fmaf(x, 1.0001f, 0.5f);
        else if (KIND == 1) x = __builtin_amdgcn_exp2f(x);
        else if (KIND == 2) x = __builtin_amdgcn_rcpf(x);
        else if (KIND == 3) {
          f32x2 v = {f[(2 * n) & 7], f[(2 * n + 1) & 7]};
          v = __builtin_elementwise_fma(v, f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
          f[(2 * n) & 7] = v.x; f[(2 * n + 1) & 7] = v.y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; i++) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int KIND, int NF, bool DEP>
void run(const char *kind, float *d_o, unsigned long long *d_c)
{
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, NF, DEP>), dim3(256), dim3(256), 0, 0, d_o, d_c, iters, 0.5f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(1024);
  hipMemcpy(c.data(), d_c, 8192, hipMemcpyDeviceToHost);
  printf("%-10s x%d per MFMA, %s accumulators: %6.1f cycles per MFMA\n", kind, NF, DEP ? "ONE (dependent chain)" : "four independent", (double)c[512] / iters / 8);
}

int main()
{
  float *d_o; unsigned long long *d_c;
  hipMalloc(&d_o, 256 * 256 * 4); hipMalloc(&d_c, 8192);
  run<0, 0, false>("none", d_o, d_c); run<0, 0, true>("none", d_o, d_c);
  run<0, 1, false>("v_fma", d_o, d_c); run<0, 2, false>("v_fma", d_o, d_c); run<0, 4, false>("v_fma", d_o, d_c); run<0, 6, false>("v_fma", d_o, d_c);
  run<0, 2, true>("v_fma", d_o, d_c); run<0, 4, true>("v_fma", d_o, d_c);
  run<1, 1, false>("v_exp", d_o, d_c); run<1, 2, false>("v_exp", d_o, d_c); run<1, 3, false>("v_exp", d_o, d_c); run<1, 4, false>("v_exp", d_o, d_c);
  run<1, 2, true>("v_exp", d_o, d_c);
  run<2, 1, false>("v_rcp", d_o, d_c); run<2, 2, false>("v_rcp", d_o, d_c); run<2, 4, false>("v_rcp", d_o, d_c);
  run<3, 1, false>("v_pk_fma", d_o, d_c); run<3, 2, false>("v_pk_fma", d_o, d_c); run<3, 3, false>("v_pk_fma", d_o, d_c);
  return 0;
}
