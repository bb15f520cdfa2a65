// clock_probe.hip -- sustained shader clock under a dense f32-MFMA load (MI355X_MICROARCH.md, DVFS give-back (6)):
// clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, stamped around the loop, median over workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o gpurun_out/clock_probe && gpurun_out/clock_probe
// argv: waves per SIMD filled (fraction of the chip = blocks / (4 * CUs)), iterations, mfma|valu
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, int iters, float seed)
{
  f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
  float a = seed + threadIdx.x * 1e-3f, b = 1.0f - seed;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (MODE == 0) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
      else {
#pragma unroll
        for (int r = 0; r < 8; r++) acc[q][r & 3] = fmaf(acc[q][r & 3], a, b);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) out[0] = 0;  // keep the loop
}

int main(int argc, char **argv)
{
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  const bool valu = argc > 3 && !strcmp(argv[3], "valu");
  unsigned long long *d;
  hipMalloc(&d, sizeof(unsigned long long) * 2 * blocks);
  std::vector<unsigned long long> h(2 * blocks);
  for (int rep = 0; rep < 4; rep++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int l = 0; l < 20; l++) {
      if (valu) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.5f);
      else hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.5f);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks);
    for (int i = 0; i < blocks; i++) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    std::sort(ghz.begin(), ghz.end());
    printf("%s blocks %d iters %d: %.3f ms per launch, cycles/iter %.1f, clock median %.3f GHz (min %.3f max %.3f)\n",
           valu ? "valu" : "mfma", blocks, iters, ms / 20, (double)h[0] / iters, ghz[blocks / 2], ghz[0], ghz[blocks - 1]);
  }
  return 0;
}
