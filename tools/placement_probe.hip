// placement_probe.hip -- where does the dispatcher put the workgroups of a launch?  Every wave records its
// (XCC, SE, CU, SIMD) from the hardware-id registers while all waves of the launch are resident (each spins
// for a fixed number of clock ticks), and the host prints the histogram of waves per SIMD and per CU.
//   placement_probe <blocks> <threads> <vgprs: 64|128|200> <lds bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

template <int NREG>
__global__ void probe(unsigned *out, long long spin, float seed)
{
  extern __shared__ float lds[];
  float r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; i++) r[i] = seed * (float)(i + threadIdx.x);
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) {
#pragma unroll
    for (int i = 0; i < NREG; i++) r[i] = fmaf(r[i], 1.0001f, 0.5f);
  }
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NREG; i++) s += r[i];
  if (threadIdx.x == 0) lds[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    out[2 * w] = hw;
    out[2 * w + 1] = xcc | (s == 12345.0f ? 0x80000000u : 0u);
  }
}

int main(int argc, char **argv)
{
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 64;
  const int regs = argc > 3 ? atoi(argv[3]) : 64, lds = argc > 4 ? atoi(argv[4]) : 16;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d  launch: %d blocks x %d threads, ~%d VGPRs, %d B LDS\n", p.gcnArchName, p.multiProcessorCount, blocks, threads, regs, lds);
  const int waves = blocks * (threads / 64);
  unsigned *d;
  hipMalloc(&d, 8 * waves);
  for (int rep = 0; rep < 2; rep++) {
    hipMemset(d, 0xff, 8 * waves);
    const long long spin = 400000;
    if (regs <= 64) hipLaunchKernelGGL(probe<40>, dim3(blocks), dim3(threads), lds, 0, d, spin, 0.5f);
    else if (regs <= 128) hipLaunchKernelGGL(probe<100>, dim3(blocks), dim3(threads), lds, 0, d, spin, 0.5f);
    else hipLaunchKernelGGL(probe<180>, dim3(blocks), dim3(threads), lds, 0, d, spin, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * waves);
    hipMemcpy(h.data(), d, 8 * waves, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu;
    for (int w = 0; w < waves; w++) {
      const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      per_cu[cukey]++;
      per_simd[(cukey << 2) | simd]++;
    }
    std::map<int, int> hs, hc;
    for (auto &kv : per_simd) hs[kv.second]++;
    for (auto &kv : per_cu) hc[kv.second]++;
    printf("rep %d: distinct CUs used %zu, distinct SIMDs used %zu | waves per SIMD:", rep, per_cu.size(), per_simd.size());
    for (auto &kv : hs) printf("  %d waves x %d SIMDs", kv.first, kv.second);
    printf(" | waves per CU:");
    for (auto &kv : hc) printf("  %d x %d CUs", kv.first, kv.second);
    printf("\n");
  }
  return 0;
}
